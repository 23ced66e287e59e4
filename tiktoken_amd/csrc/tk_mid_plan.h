// Where a document of 2 .. 128 KiB is cut into segments for the small kernel (encode_mid, tk_api.hip).  Host-only and free of HIP: the CPU tests
// compile it into their host build of the device logic and drive it from there (tks_mid_plan).
//
// A cut may only stand where a piece starts whatever is on either side, so that the segments' tokens put together ARE the document's tokens
// (reference src/lib.rs:236-249: the pieces of the regex, each encoded on its own).  The one place used: an ASCII letter followed by a space
// -- for a pattern whose table says that a space after a letter always starts a piece (tk_core::mid_cut; true of the stock families:
// ` ?\p{L}+` ends at the space, and no alternative continues a letter across it).
#pragma once
#include <stdint.h>
#include "tk_common.h"

#define TK_MID_SEGMENT_MAX 2048u  // bytes of a segment at most (= TK_SMALL_MAX, tk_fused.h: what one workgroup of tk_k_small takes)
#define TK_MID_SEGMENTS 64u       // most segments a document is PLANNED in; documents of up to TK_MID_SEGMENT_MAX * TK_MID_SEGMENTS bytes come here
#define TK_SMALL_SLOTS 72u        // small calls in flight at the same time (tk_core::SmallSlot; = TK_SMALL_BATCH: one launch can carry them all) --
                                  // a few more than TK_MID_SEGMENTS: segments end a word short of their limit

// Is "letter, then space" a certain piece start of this pattern?  Both cases of letter, in the table that travels with the pattern
// (TkTables::cert: the family's static table for a stock pattern, what tk_pattern.cpp derived for a custom one of the families).
inline bool tk_mid_cut_certain(const uint16_t* cert) { return ((cert[TK_C_LL] >> TK_C_SP) & 1u) && ((cert[TK_C_LU] >> TK_C_SP) & 1u); }

// cuts[0] = 0 < cuts[1] < ... < cuts[k] = n; returns k (>= 2), or 0 with *reason set: the general pipeline takes the document.
// About equal segments, as many as there are slots for, at least 1 KiB each (a segment costs what its pieces cost one after the other:
// shorter segments, shorter call).
inline uint32_t tk_mid_plan(const uint8_t* utf8, uint32_t n, uint32_t* cuts /* [TK_SMALL_SLOTS + 1] */, const char** reason) {
    uint32_t want = (n + 1023u) / 1024u;
    if (want > TK_MID_SEGMENTS) want = TK_MID_SEGMENTS;
    if (want == 0u) want = 1u;
    const uint32_t target = (n + want - 1u) / want;
    uint32_t k = 0, pos = 0;
    cuts[0] = 0;
    while (n - pos > TK_MID_SEGMENT_MAX || (k + 1u < want && n - pos > target + target / 2u)) {
        if (k + 1u >= TK_SMALL_SLOTS) return *reason = "more segments than slots", 0u;
        uint32_t reach = target + target / 4u;
        if (reach > TK_MID_SEGMENT_MAX) reach = TK_MID_SEGMENT_MAX;
        const uint32_t hi = pos + reach;
        const uint32_t lo = pos + 64u;  // (the search runs down from hi and takes the first cut it meets: the floor only matters in text with few cuts)
        uint32_t cut = 0;
        for (uint32_t i = hi < n - 1u ? hi : n - 1u; i > lo; --i) {
            const uint8_t p = utf8[i - 1];
            if (utf8[i] == ' ' && ((p >= 'a' && p <= 'z') || (p >= 'A' && p <= 'Z'))) {
                cut = i;
                break;
            }
        }
        if (!cut) return *reason = "no cut in a window", 0u;
        cuts[++k] = pos = cut;
    }
    cuts[++k] = n;  // k segments
    if (k < 2u) return *reason = "one segment", 0u;
    return k;
}
