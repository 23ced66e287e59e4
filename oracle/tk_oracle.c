/* See tk_oracle.h.  CPU parity oracle -- TEST INFRASTRUCTURE ONLY, never linked by the product. */
#define _GNU_SOURCE
#include "tk_oracle.h"

#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "tk_unicode_tables.inc"

/* ------------------------------------------------------------------ classes */
enum { C_CONT = 0, C_NL, C_SP, C_WSO, C_LU, C_LL, C_LC, C_MK, C_NU, C_AP, C_SL, C_OT, C_END };
#define B(c) (1u << (c))
#define M_WS (B(C_NL) | B(C_SP) | B(C_WSO))
#define M_L (B(C_LU) | B(C_LL) | B(C_LC))
#define M_OTHER (B(C_MK) | B(C_AP) | B(C_SL) | B(C_OT)) /* [^\s\p{L}\p{N}] */
#define M_WORD (M_L | B(C_MK))                         /* o200k letter-ish */
#define M_UPPERISH (B(C_LU) | B(C_LC) | B(C_MK))       /* [\p{Lu}\p{Lt}\p{Lm}\p{Lo}\p{M}] */
#define M_LOWERISH (B(C_LL) | B(C_LC) | B(C_MK))       /* [\p{Ll}\p{Lm}\p{Lo}\p{M}] */

static inline int cls_of_cp(uint32_t cp) {
    if (cp > 0x10FFFF) return C_OT;
    return tk_uc_stage2[(uint32_t)tk_uc_stage1[cp >> 8] * 256u + (cp & 255u)];
}

/* class and byte length of the char starting at t[i]; C_END (length 0) at i >= end */
static inline int cls_at(const uint8_t* t, uint64_t i, uint64_t end, int* clen) {
    if (i >= end) {
        *clen = 0;
        return C_END;
    }
    uint8_t b = t[i];
    if (b < 0x80) {
        *clen = 1;
        return cls_of_cp(b);
    }
    int n = b >= 0xF0 ? 4 : (b >= 0xE0 ? 3 : 2);
    if (i + (uint64_t)n > end) { /* truncated sequence: cannot happen for valid UTF-8 */
        *clen = (int)(end - i);
        return C_OT;
    }
    uint32_t cp;
    if (n == 2)
        cp = ((uint32_t)(b & 0x1F) << 6) | (t[i + 1] & 0x3F);
    else if (n == 3)
        cp = ((uint32_t)(b & 0x0F) << 12) | ((uint32_t)(t[i + 1] & 0x3F) << 6) | (t[i + 2] & 0x3F);
    else
        cp = ((uint32_t)(b & 0x07) << 18) | ((uint32_t)(t[i + 1] & 0x3F) << 12) |
             ((uint32_t)(t[i + 2] & 0x3F) << 6) | (t[i + 3] & 0x3F);
    *clen = n;
    return cls_of_cp(cp);
}

static inline uint64_t run_end(const uint8_t* t, uint64_t s, uint64_t end, uint32_t mask) {
    int n;
    while (s < end && (B(cls_at(t, s, end, &n)) & mask)) s += (uint64_t)n;
    return s;
}

/* byte length of a contraction at t[p]=='\'' (openai_public.py:13 case-sensitive; :89,:106-107
 * case-insensitive, where U+017F folds to 's'), or 0 */
static int contraction_len(const uint8_t* t, uint64_t p, uint64_t end, int ci) {
    if (p + 1 >= end) return 0;
    uint8_t a = t[p + 1];
    uint8_t b = p + 2 < end ? t[p + 2] : 0;
    if (ci) {
        if (a == 0xC5 && b == 0xBF) return 3; /* long s */
        uint8_t al = (uint8_t)(a | 0x20), bl = (uint8_t)(b | 0x20);
        int a_alpha = (al >= 'a' && al <= 'z'), b_alpha = (bl >= 'a' && bl <= 'z');
        if (!a_alpha) return 0;
        if (al == 's' || al == 'd' || al == 'm' || al == 't') return 2;
        if (!b_alpha) return 0;
        if ((al == 'l' && bl == 'l') || (al == 'v' && bl == 'e') || (al == 'r' && bl == 'e')) return 3;
        return 0;
    }
    if (a == 's' || a == 'd' || a == 'm' || a == 't') return 2;
    if ((a == 'l' && b == 'l') || (a == 'v' && b == 'e') || (a == 'r' && b == 'e')) return 3;
    return 0;
}

static uint64_t ws_tail(const uint8_t* t, uint64_t p, uint64_t end, int pat) {
    uint64_t q = p, last_start = p, after_last_nl = 0;
    int nchars = 0, has_nl = 0, n;
    for (;;) {
        int c = cls_at(t, q, end, &n);
        if (!(B(c) & M_WS)) break;
        last_start = q;
        q += (uint64_t)n;
        nchars++;
        if (c == C_NL) {
            has_nl = 1;
            after_last_nl = q;
        }
    }
    if (pat != TKO_PAT_O200K && q == end) return q; /* \s++$ */
    if (pat != TKO_PAT_R50K && has_nl) return after_last_nl; /* \s*[\r\n] / \s*[\r\n]+ */
    if (q == end) return q;                                  /* \s+(?!\S) at end of haystack */
    if (nchars >= 2) return last_start;                      /* \s+(?!\S) backs off one char */
    return q;                                                /* \s */
}

static uint64_t o200k_word(const uint8_t* t, uint64_t s, uint64_t end) {
    uint64_t r_end = s, after_last_c = 0;
    int has_c = 0, n;
    for (;;) {
        int c = cls_at(t, r_end, end, &n);
        if (!(B(c) & M_UPPERISH)) break;
        r_end += (uint64_t)n;
        if (c != C_LU) {
            has_c = 1;
            after_last_c = r_end;
        }
    }
    uint64_t t_end = run_end(t, r_end, end, M_LOWERISH);
    uint64_t e;
    if (t_end > r_end)
        e = t_end;
    else if (has_c)
        e = after_last_c;
    else if (r_end > s)
        e = r_end;
    else
        return 0; /* unreachable when called on a letter-ish start */
    if (e < end && t[e] == '\'') e += (uint64_t)contraction_len(t, e, end, 1);
    return e;
}

static uint64_t piece_end(const uint8_t* t, uint64_t p, uint64_t end, int pat) {
    int n0, n1, ns;
    int c = cls_at(t, p, end, &n0);
    int nxt = cls_at(t, p + (uint64_t)n0, end, &n1);
    if (pat == TKO_PAT_R50K) {
        if (c == C_AP) {
            int k = contraction_len(t, p, end, 0);
            if (k) return p + (uint64_t)k;
        }
        uint64_t s = (c == C_SP && nxt != C_END) ? p + 1 : p;
        int k = cls_at(t, s, end, &ns);
        if (B(k) & M_L) return run_end(t, s, end, M_L);
        if (k == C_NU) return run_end(t, s, end, B(C_NU));
        if (B(k) & M_OTHER) return run_end(t, s, end, M_OTHER);
        return ws_tail(t, p, end, pat);
    }
    if (pat == TKO_PAT_CL100K) {
        if (c == C_AP) {
            int k = contraction_len(t, p, end, 1);
            if (k) return p + (uint64_t)k;
        }
        if ((B(c) & M_L) || (c != C_NL && c != C_NU && (B(nxt) & M_L))) return run_end(t, p + (uint64_t)n0, end, M_L);
        if (c == C_NU) {
            uint64_t e = p;
            for (int i = 0; i < 3; ++i) {
                int n, k = cls_at(t, e, end, &n);
                if (k != C_NU) break;
                e += (uint64_t)n;
            }
            return e;
        }
        uint64_t s = (c == C_SP && nxt != C_END) ? p + 1 : p;
        int k = cls_at(t, s, end, &ns);
        if (B(k) & M_OTHER) {
            uint64_t e = run_end(t, s, end, M_OTHER);
            return run_end(t, e, end, B(C_NL));
        }
        return ws_tail(t, p, end, pat);
    }
    /* o200k */
    if (B(c) & M_WORD) return o200k_word(t, p, end);
    if (c != C_NL && c != C_NU && (B(nxt) & M_WORD)) return o200k_word(t, p + (uint64_t)n0, end);
    if (c == C_NU) {
        uint64_t e = p;
        for (int i = 0; i < 3; ++i) {
            int n, k = cls_at(t, e, end, &n);
            if (k != C_NU) break;
            e += (uint64_t)n;
        }
        return e;
    }
    uint64_t s = (c == C_SP && nxt != C_END) ? p + 1 : p;
    int k = cls_at(t, s, end, &ns);
    if (B(k) & M_OTHER) {
        uint64_t e = run_end(t, s, end, M_OTHER);
        return run_end(t, e, end, B(C_NL) | B(C_SL));
    }
    return ws_tail(t, p, end, pat);
}

int64_t tko_split(int pattern, const uint8_t* text, uint64_t len, uint64_t* piece_ends, uint64_t cap) {
    uint64_t p = 0, n = 0;
    while (p < len) {
        uint64_t e = piece_end(text, p, len, pattern);
        if (e <= p) e = p + 1; /* defensive: never loop */
        if (n < cap) piece_ends[n] = e;
        n++;
        p = e;
    }
    return n <= cap ? (int64_t)n : -(int64_t)n;
}

/* ------------------------------------------------------------------ vocabulary (bytes -> rank) */
#define RANK_MAX 0xFFFFFFFFu

typedef struct {
    const uint8_t* key;
    uint32_t len;
    uint32_t rank;
} slot_t;

struct tko_vocab {
    int pattern;
    uint8_t* blob; /* owned copy of all key bytes */
    slot_t* slots;
    uint64_t mask;
    /* specials */
    uint8_t* sblob;
    uint64_t* soff;
    uint32_t* sids;
    uint64_t ns;
};

static inline uint64_t hash_bytes(const uint8_t* p, uint64_t n) {
    uint64_t h = 0xCBF29CE484222325ull ^ (n * 0x9E3779B97F4A7C15ull);
    while (n >= 8) {
        uint64_t w;
        memcpy(&w, p, 8);
        h = (h ^ w) * 0x9FB21C651E98DF25ull;
        h ^= h >> 29;
        p += 8;
        n -= 8;
    }
    uint64_t w = 0;
    memcpy(&w, p, n);
    h = (h ^ w) * 0x9FB21C651E98DF25ull;
    h ^= h >> 32;
    return h;
}

static inline uint32_t vget(const tko_vocab* v, const uint8_t* p, uint64_t n) {
    uint64_t i = hash_bytes(p, n) & v->mask;
    for (;;) {
        const slot_t* s = &v->slots[i];
        if (!s->key) return RANK_MAX;
        if (s->len == n && memcmp(s->key, p, n) == 0) return s->rank;
        i = (i + 1) & v->mask;
    }
}

tko_vocab* tko_vocab_new(const uint8_t* blob, const uint64_t* off, const uint32_t* ids, uint64_t n,
                         const uint8_t* sblob, const uint64_t* soff, const uint32_t* sids, uint64_t ns,
                         int pattern) {
    tko_vocab* v = (tko_vocab*)calloc(1, sizeof *v);
    v->pattern = pattern;
    uint64_t cap = 16;
    while (cap < 2 * n + 2) cap <<= 1;
    v->mask = cap - 1;
    v->slots = (slot_t*)calloc(cap, sizeof(slot_t));
    v->blob = (uint8_t*)malloc(off[n] ? off[n] : 1);
    memcpy(v->blob, blob, off[n]);
    for (uint64_t k = 0; k < n; ++k) {
        const uint8_t* key = v->blob + off[k];
        uint64_t len = off[k + 1] - off[k];
        uint64_t i = hash_bytes(key, len) & v->mask;
        while (v->slots[i].key) i = (i + 1) & v->mask;
        v->slots[i].key = key;
        v->slots[i].len = (uint32_t)len;
        v->slots[i].rank = ids[k];
    }
    v->ns = ns;
    v->sblob = (uint8_t*)malloc(ns && soff[ns] ? soff[ns] : 1);
    v->soff = (uint64_t*)malloc((ns + 1) * sizeof(uint64_t));
    v->sids = (uint32_t*)malloc((ns ? ns : 1) * sizeof(uint32_t));
    if (ns) {
        memcpy(v->sblob, sblob, soff[ns]);
        memcpy(v->soff, soff, (ns + 1) * sizeof(uint64_t));
        memcpy(v->sids, sids, ns * sizeof(uint32_t));
    } else {
        v->soff[0] = 0;
    }
    return v;
}

void tko_vocab_free(tko_vocab* v) {
    if (!v) return;
    free(v->slots);
    free(v->blob);
    free(v->sblob);
    free(v->soff);
    free(v->sids);
    free(v);
}

/* ------------------------------------------------------------------ byte_pair_merge (lib.rs:140-196) */
typedef struct {
    uint32_t start;
    uint32_t rank;
} part_t;

static int64_t tko_bpe_small(const tko_vocab* v, const uint8_t* piece, uint32_t n, uint32_t* out) {
    part_t stackbuf[128];
    part_t* parts = stackbuf; /* n < 100 here */
    uint32_t np = 0;
    uint32_t min_rank = RANK_MAX, min_i = 0;
    for (uint32_t i = 0; i + 1 < n; ++i) {
        uint32_t r = vget(v, piece + i, 2);
        if (r < min_rank) {
            min_rank = r;
            min_i = i;
        }
        parts[np].start = i;
        parts[np++].rank = r;
    }
    parts[np].start = n - 1;
    parts[np++].rank = RANK_MAX;
    parts[np].start = n;
    parts[np++].rank = RANK_MAX;
    while (min_rank != RANK_MAX) {
        uint32_t i = min_i;
        /* get_rank(i) spans parts[i].start .. parts[i+3].start because parts[i+1] is not yet removed */
        if (i > 0)
            parts[i - 1].rank = (i - 1 + 3 < np)
                                    ? vget(v, piece + parts[i - 1].start, parts[i + 2].start - parts[i - 1].start)
                                    : RANK_MAX;
        parts[i].rank = (i + 3 < np) ? vget(v, piece + parts[i].start, parts[i + 3].start - parts[i].start) : RANK_MAX;
        memmove(&parts[i + 1], &parts[i + 2], (np - (i + 2)) * sizeof(part_t));
        np--;
        min_rank = RANK_MAX;
        for (uint32_t j = 0; j + 1 < np; ++j)
            if (parts[j].rank < min_rank) {
                min_rank = parts[j].rank;
                min_i = j;
            }
    }
    for (uint32_t k = 0; k + 1 < np; ++k) out[k] = vget(v, piece + parts[k].start, parts[k + 1].start - parts[k].start);
    return (int64_t)np - 1;
}

/* ------------------------------------------------------------------ heap variant (lib.rs:47-138) */
typedef struct {
    uint32_t rank;
    uint32_t start;
} merge_t;
typedef struct {
    uint32_t prev, end, next_end, next_rank, cur_rank;
} state_t;
typedef struct {
    merge_t* a;
    uint64_t n, cap;
} heap_t;

static inline int merge_less(merge_t x, merge_t y) { return x.rank < y.rank || (x.rank == y.rank && x.start < y.start); }
static void heap_push(heap_t* h, merge_t m) {
    if (h->n == h->cap) {
        h->cap = h->cap ? h->cap * 2 : 64;
        h->a = (merge_t*)realloc(h->a, h->cap * sizeof(merge_t));
    }
    uint64_t i = h->n++;
    while (i > 0) {
        uint64_t p = (i - 1) / 2;
        if (!merge_less(m, h->a[p])) break;
        h->a[i] = h->a[p];
        i = p;
    }
    h->a[i] = m;
}
static merge_t heap_pop(heap_t* h) {
    merge_t top = h->a[0];
    merge_t m = h->a[--h->n];
    uint64_t i = 0;
    for (;;) {
        uint64_t c = 2 * i + 1;
        if (c >= h->n) break;
        if (c + 1 < h->n && merge_less(h->a[c + 1], h->a[c])) c++;
        if (!merge_less(h->a[c], m)) break;
        h->a[i] = h->a[c];
        i = c;
    }
    if (h->n) h->a[i] = m;
    return top;
}

static void potential_merge(const tko_vocab* v, const uint8_t* piece, uint32_t n, state_t* st, heap_t* h,
                            uint32_t start, uint32_t next_end_item) {
    st[start].next_end = next_end_item;
    st[start].next_rank = RANK_MAX;
    if (next_end_item <= n) {
        uint32_t r = vget(v, piece + start, next_end_item - start);
        if (r != RANK_MAX) {
            merge_t m = {r, start};
            heap_push(h, m);
            st[start].next_rank = r;
        }
    }
}

static int64_t tko_bpe_large(const tko_vocab* v, const uint8_t* piece, uint32_t n, uint32_t* out) {
    state_t* st = (state_t*)malloc((uint64_t)n * sizeof(state_t));
    heap_t h = {0, 0, 0};
    st[0].prev = RANK_MAX;
    st[0].end = 1;
    st[0].next_end = 2;
    st[0].next_rank = RANK_MAX;
    st[0].cur_rank = RANK_MAX;
    for (uint32_t i = 0; i + 1 < n; ++i) {
        uint32_t r = vget(v, piece + i, 2);
        if (r != RANK_MAX) {
            merge_t m = {r, i};
            heap_push(&h, m);
            st[i].next_rank = r;
        }
        st[i + 1].prev = i;
        st[i + 1].end = i + 2;
        st[i + 1].next_end = i + 3;
        st[i + 1].next_rank = RANK_MAX;
        st[i + 1].cur_rank = RANK_MAX;
    }
    while (h.n) {
        merge_t left = heap_pop(&h);
        if (left.rank == RANK_MAX) break;
        if (left.rank != st[left.start].next_rank) continue; /* invalidated */
        uint32_t ls = left.start;
        uint32_t right_start = st[ls].end;
        uint32_t right_end = st[ls].next_end;
        uint32_t right_next_end = st[right_start].next_end;
        st[ls].cur_rank = st[ls].next_rank;
        st[ls].end = right_end;
        potential_merge(v, piece, n, st, &h, ls, right_next_end);
        if (right_end < n) st[right_end].prev = ls;
        if (ls > 0) potential_merge(v, piece, n, st, &h, st[ls].prev, right_end);
        st[right_start].next_rank = RANK_MAX;
    }
    int64_t k = 0;
    for (uint32_t i = 0; i < n; i = st[i].end)
        out[k++] = st[i].cur_rank != RANK_MAX ? st[i].cur_rank : vget(v, piece + i, st[i].end - i);
    free(st);
    free(h.a);
    return k;
}

/* byte_pair_encode (lib.rs:198-211) behind the whole-piece probe (lib.rs:367-369) */
int64_t tko_encode_piece(const tko_vocab* v, const uint8_t* piece, uint64_t len, uint32_t* out, uint64_t cap) {
    if (len == 0) return 0;
    if (cap < len) return -1;
    uint32_t r = vget(v, piece, len);
    if (r != RANK_MAX) {
        out[0] = r;
        return 1;
    }
    if (len == 1) { /* reference would panic (lib.rs:202): every single byte must be in the vocabulary */
        out[0] = RANK_MAX;
        return 1;
    }
    if (len < 100) return tko_bpe_small(v, piece, (uint32_t)len, out);
    return tko_bpe_large(v, piece, (uint32_t)len, out);
}

int64_t tko_encode_ordinary(const tko_vocab* v, const uint8_t* text, uint64_t len, uint32_t* out, uint64_t cap) {
    if (cap < len) return -1;
    uint64_t p = 0;
    int64_t n = 0;
    while (p < len) {
        uint64_t e = piece_end(text, p, len, v->pattern);
        if (e <= p) e = p + 1;
        n += tko_encode_piece(v, text + p, e - p, out + n, cap - (uint64_t)n);
        p = e;
    }
    return n;
}

/* leftmost allowed special at or after `from`; on ties the longest (see oracle/py_oracle.py) */
static int find_special(const tko_vocab* v, const uint8_t* text, uint64_t len, uint64_t from, const uint8_t* allowed,
                        uint64_t* pos_out, uint64_t* len_out, uint32_t* id_out) {
    int found = 0;
    uint64_t best = 0, blen = 0;
    uint32_t bid = 0;
    for (uint64_t k = 0; k < v->ns; ++k) {
        if (!allowed[k]) continue;
        uint64_t sl = v->soff[k + 1] - v->soff[k];
        if (sl == 0 || from + sl > len) continue;
        uint64_t hl = len - from;
        if (found && best - from + sl < hl) hl = best - from + sl; /* only matches starting at <= best matter */
        const uint8_t* hit = (const uint8_t*)memmem(text + from, hl, v->sblob + v->soff[k], sl);
        if (!hit) continue;
        uint64_t pos = (uint64_t)(hit - text);
        if (!found || pos < best || (pos == best && sl > blen)) {
            found = 1;
            best = pos;
            blen = sl;
            bid = v->sids[k];
        }
    }
    if (found) {
        *pos_out = best;
        *len_out = blen;
        *id_out = bid;
    }
    return found;
}

int64_t tko_encode(const tko_vocab* v, const uint8_t* text, uint64_t len, const uint32_t* allowed_ids,
                   uint64_t n_allowed, uint32_t* out, uint64_t cap) {
    if (cap < len) return -1;
    uint8_t* allowed = (uint8_t*)calloc(v->ns ? v->ns : 1, 1);
    int any = 0;
    for (uint64_t k = 0; k < v->ns; ++k)
        for (uint64_t j = 0; j < n_allowed; ++j)
            if (v->sids[k] == allowed_ids[j]) {
                allowed[k] = 1;
                any = 1;
            }
    uint64_t start = 0;
    int64_t n = 0;
    for (;;) {
        uint64_t pos = 0, sl = 0;
        uint32_t id = 0;
        int hit = any ? find_special(v, text, len, start, allowed, &pos, &sl, &id) : 0;
        uint64_t end = hit ? pos : len;
        /* the slice text[start..end) is an independent haystack (lib.rs:402-405) */
        n += tko_encode_ordinary(v, text + start, end - start, out + n, cap - (uint64_t)n);
        if (!hit) break;
        out[n++] = id;
        start = pos + sl;
    }
    free(allowed);
    return n;
}

/* ------------------------------------------------------------------ batch (core.py:164-206) */
typedef struct {
    const tko_vocab* v;
    const uint8_t* blob;
    const uint64_t* doc_off;
    uint64_t n_docs;
    int mode;
    const uint32_t* allowed_ids;
    uint64_t n_allowed;
    uint32_t* tokens;
    uint64_t* counts;
    int tid, nth;
    uint64_t* next; /* shared: the next block of documents nobody has taken yet */
} job_t;

static void* worker(void* arg) {
    job_t* j = (job_t*)arg;
    /* contiguous blocks of 16 documents, handed out as threads become free (a thread pool over documents, core.py:175) */
    for (;;) {
        uint64_t base = __atomic_fetch_add(j->next, 1, __ATOMIC_RELAXED) * 16;
        if (base >= j->n_docs) break;
        uint64_t hi = base + 16 < j->n_docs ? base + 16 : j->n_docs;
        for (uint64_t d = base; d < hi; ++d) {
            uint64_t a = j->doc_off[d], b = j->doc_off[d + 1];
            int64_t n = j->mode == 0
                            ? tko_encode_ordinary(j->v, j->blob + a, b - a, j->tokens + a, b - a)
                            : tko_encode(j->v, j->blob + a, b - a, j->allowed_ids, j->n_allowed, j->tokens + a, b - a);
            j->counts[d] = (uint64_t)n;
        }
    }
    return 0;
}

static double g_last_encode_seconds = 0.0;
/* wall time of the parallel per-document encode of the last tko_encode_batch call (what the reference's thread pool does,
 * core.py:175); the packing of the results into one buffer that follows is this oracle's own addition and is excluded */
double tko_last_encode_seconds(void) { return g_last_encode_seconds; }
static double now_seconds(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

int tko_encode_batch(const tko_vocab* v, const uint8_t* blob, const uint64_t* doc_off, uint64_t n_docs, int mode,
                     const uint32_t* allowed_ids, uint64_t n_allowed, int n_threads, uint32_t* tokens_out,
                     uint64_t* tok_off_out) {
    const double t_start = now_seconds();
    if (n_threads < 1) n_threads = 1;
    if (n_threads > 256) n_threads = 256;
    uint64_t* counts = (uint64_t*)calloc(n_docs ? n_docs : 1, sizeof(uint64_t));
    pthread_t th[256];
    job_t jobs[256];
    uint64_t next = 0;
    for (int t = 0; t < n_threads; ++t) {
        job_t j = {v, blob, doc_off, n_docs, mode, allowed_ids, n_allowed, tokens_out, counts, t, n_threads, &next};
        jobs[t] = j;
        if (n_threads > 1) pthread_create(&th[t], 0, worker, &jobs[t]);
    }
    if (n_threads == 1)
        worker(&jobs[0]);
    else
        for (int t = 0; t < n_threads; ++t) pthread_join(th[t], 0);
    g_last_encode_seconds = now_seconds() - t_start;
    /* tokens of document d sit at index doc_off[d]; compact them forward (dest <= src always) */
    uint64_t w = 0;
    for (uint64_t d = 0; d < n_docs; ++d) {
        tok_off_out[d] = w;
        memmove(tokens_out + w, tokens_out + doc_off[d], counts[d] * sizeof(uint32_t));
        w += counts[d];
    }
    tok_off_out[n_docs] = w;
    free(counts);
    return 0;
}
