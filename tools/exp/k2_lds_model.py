"""Scratch: LDS bank-conflict model of the IDCT/colour kernel (4:2:0) after the lane groups of the CDNA4 guide: extra LDS cycles per
wavefront and strip for a list of 24 block offsets (words) and a row pitch.  python tools/exp/k2_lds_model.py prints the shipped layout."""
import itertools, random
ZZ=[0,1,8,16,9,2,3,10,17,24,32,25,18,11,4,5,12,19,26,33,40,48,41,34,27,20,13,6,7,14,21,28,35,42,49,56,57,50,43,36,29,22,15,23,30,37,44,51,58,59,52,45,38,31,39,46,53,60,61,54,47,55,62,63]
G128=[list(range(0,4))+list(range(12,16))+list(range(20,28)), list(range(4,12))+list(range(16,20))+list(range(28,32))]
G128=G128+[[x+32 for x in g] for g in G128]
G32=[list(range(0,32)),list(range(32,64))]
GW128=[list(range(8*i,8*i+8)) for i in range(8)]
def cost(groups, addr, nbanks, width):
    # addr: list of 64 word addresses (None = inactive); each access covers `width` banks from addr%nbanks
    tot=0
    for g in groups:
        banks={}
        for l in g:
            a=addr[l]
            if a is None: continue
            for w in range(width):
                banks.setdefault((a+w)%nbanks,set()).add(a)
        tot+=max([len(s) for s in banks.values()]+[1])
    return tot
def evaluate(boff, rowpitch=8, detail=False):
    # boff: list of 24 block offsets (words); intra-block: row*rowpitch+col
    res={}
    def nat2w(n): return (n//8)*rowpitch+(n%8)
    c=0
    # A1 scatter ds_write_b32: bank%32, 2-way free -> cost max(mult,2)/2? model: array cycles = mult per group, instruction takes >= 2 array cycles... use extra = max(0, mult-2) per group
    a1=0
    for i in range(3):
        for e in range(8):
            addr=[boff[(l>>3)+8*i]+nat2w(ZZ[8*(l&7)+e]) for l in range(64)]
            for g in G32:
                banks={}
                for l in g: banks.setdefault(addr[l]%32,set()).add(addr[l])
                m=max(len(s) for s in banks.values())
                a1+=max(0,m-2)   # transfer hides 2-way
    res['scatter']=a1
    # A2 column pass reads (b32, %32) and writes
    a2r=a2w=0
    for i in range(3):
        for k in range(8):
            addr=[boff[(l>>3)+8*i]+k*rowpitch+(l&7) for l in range(64)]
            a2r+=cost(G32,addr,32,1)-2
            for g in G32:
                banks={}
                for l in g: banks.setdefault(addr[l]%32,set()).add(addr[l])
                a2w+=max(0,max(len(s) for s in banks.values())-2)
    res['col_r']=a2r; res['col_w']=a2w
    # A3 chroma rows
    a3r=a3w=0
    for half in (0,4):
        addr=[boff[((l>>3)>>1)*6+4+((l>>3)&1)]+(l&7)*rowpitch+half for l in range(64)]
        a3r+=cost(G128,addr,64,4)-4
        a3w+=cost(GW128,addr,32,4)-8
    res['chroma_r']=a3r; res['chroma_w']=a3w
    # A4 phase 3
    a4=0
    for u in range(2):
        cb=[boff[(l>>4)*6+4]+((l&15)>>1)*rowpitch+u*4 for l in range(64)]
        cr=[boff[(l>>4)*6+5]+((l&15)>>1)*rowpitch+u*4 for l in range(64)]
        a4+=cost(G128,cb,64,4)-4+cost(G128,cr,64,4)-4
        for half in (0,4):
            y=[boff[(l>>4)*6+((l&15)>>3)*2+u]+((l&15)&7)*rowpitch+half for l in range(64)]
            a4+=cost(G128,y,64,4)-4
    res['rows']=a4
    res['total']=sum(res.values())
    return res
if __name__=="__main__":
    print(evaluate([72*b for b in range(24)]))
