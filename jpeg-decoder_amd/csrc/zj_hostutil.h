// zj_hostutil.h -- host-internal helpers shared by zj_host.cpp and zj_api.hip.
#pragma once
#include "../../include/zj.h"
#include "zj_core.h"

// True if the image is inside the decode path (3 components, luma sampling 1..2, chroma 1x1,
// tables present).
bool zj_image_supported(const zj_image *im);

// Build one device lookup table (ZJ_TBL_ENTRIES entries) and its canonical description from a DHT
// table.  Returns false if the table is not a prefix code.
// l2_cap: entries of the second level that may be used (ZJ_L2_SIZE; smaller values push long codes to the
// canonical search, which is how the tests reach that path with ordinary tables)
bool zj_build_table(const zj_huff *h, uint16_t *tbl, ZjCanon *canon, uint32_t l2_cap = ZJ_L2_SIZE);
void zj_build_walk_table(const uint16_t *tbl, bool isdc, uint16_t *wtbl);

// The table in pair-entry form (zj_core.h): ZJ_PT_ENTRIES entries.  `canon` is the table's canonical
// description from zj_build_table.  Returns false when the long codes do not fit the second level; the
// first-level entries of long prefixes then send the walk to the canonical search.
bool zj_build_pair_table(const ZjCanon *canon, bool isdc, uint32_t *ptbl, const ZjCanon *partner = nullptr);

// The AC table in the direct form of its write entries (zj_core.h): ZJ_WT_ENTRIES entries and the form's parameter.
// Returns false when the table's long codes do not fit; `wt` is then undefined.
bool zj_build_direct_table(const ZjCanon *canon, uint32_t *wt, uint32_t *t16);
