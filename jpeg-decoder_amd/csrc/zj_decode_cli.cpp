// zj_decode_cli.cpp -- command-line driver, the counterpart of examples/decode.rs:1-9 (which
// hard-codes ./test-images/profile.jpg and always writes out.ppm).
//   zj_decode <in.jpg> [-o out.ppm] [--device N]
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "zj_decoder.hpp"

int main(int argc, char **argv)
{
    const char *in = nullptr, *out = "out.ppm";
    int device = 0;
    for (int i = 1; i < argc; i++) {
        if (!strcmp(argv[i], "-o") && i + 1 < argc) out = argv[++i];
        else if (!strcmp(argv[i], "--device") && i + 1 < argc) device = atoi(argv[++i]);
        else in = argv[i];
    }
    if (!in) { fprintf(stderr, "usage: zj_decode <in.jpg> [-o out.ppm] [--device N]\n"); return 2; }
    FILE *f = fopen(in, "rb");
    if (!f) { perror(in); return 1; }
    zj::Decoder dec(f, device);
    std::vector<uint8_t> rgb;
    zj::DecodeError e = dec.decode_to(rgb, out);
    if (!e.ok()) { fprintf(stderr, "decode failed: %s (status %u)\n", e.message.c_str(), e.status); return 1; }
    printf("%ux%u -> %s\n", dec.width(), dec.height(), out);
    return 0;
}
