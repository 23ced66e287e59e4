// Calls per second of tk_encode_ordinary on one core from N native threads (no interpreter in the loop: Python threads stop at ~85 k calls/s
// because every call's marshalling holds the GIL).  Built and driven by tools/small_call.py:
//   double tk_small_threads(fn tk_encode_ordinary, fn tk_free, void* core, const uint8_t* text, uint64_t len, int nthreads, int calls_per_thread)
#include <pthread.h>
#include <stdint.h>
#include <time.h>
typedef int (*enc_fn)(void*, const uint8_t*, uint64_t, uint32_t**, uint64_t*);
typedef void (*free_fn)(void*);
struct job { enc_fn enc; free_fn fr; void* core; const uint8_t* text; uint64_t len; int calls; uint64_t tokens; int bad; };
static void* work(void* p) {
    struct job* j = (struct job*)p;
    for (int i = 0; i < j->calls; ++i) {
        uint32_t* out = 0; uint64_t n = 0;
        if (j->enc(j->core, j->text, j->len, &out, &n) != 0) { j->bad++; continue; }
        j->tokens += n;
        j->fr(out);
    }
    return 0;
}
double tk_small_threads(enc_fn enc, free_fn fr, void* core, const uint8_t* text, uint64_t len, int nthreads, int calls, uint64_t* tokens_out, int* bad_out) {
    pthread_t th[256]; struct job jb[256];
    if (nthreads > 256) nthreads = 256;
    struct timespec a, b;
    clock_gettime(CLOCK_MONOTONIC, &a);
    for (int t = 0; t < nthreads; ++t) { jb[t] = (struct job){enc, fr, core, text, len, calls, 0, 0}; pthread_create(&th[t], 0, work, &jb[t]); }
    uint64_t tok = 0; int bad = 0;
    for (int t = 0; t < nthreads; ++t) { pthread_join(th[t], 0); tok += jb[t].tokens; bad += jb[t].bad; }
    clock_gettime(CLOCK_MONOTONIC, &b);
    if (tokens_out) *tokens_out = tok;
    if (bad_out) *bad_out = bad;
    return (double)nthreads * calls / ((b.tv_sec - a.tv_sec) + 1e-9 * (b.tv_nsec - a.tv_nsec));
}
