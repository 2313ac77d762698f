/* cache_sim.c -- developer tool (tools/l2_bound.py drives it): misses of ONE fully associative cache of `cap` units on a
 * reference stream of unit ids (int32, binary file), under LRU and under Belady's optimal replacement (the minimum any
 * replacement policy can reach for THIS order of references).  Prints "n lru_misses opt_misses distinct".
 *   cc -O2 -o cache_sim cache_sim.c ;  ./cache_sim stream.bin <units in universe> <capacity in units> */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

typedef struct { int64_t key; int32_t id; } Ent;  /* max-heap on key (next use) */
static Ent* heap; static int64_t hn;
static void push(int64_t key, int32_t id) {
    int64_t i = hn++; heap[i].key = key; heap[i].id = id;
    while (i > 0) { int64_t p = (i - 1) / 2; if (heap[p].key >= heap[i].key) break; Ent t = heap[p]; heap[p] = heap[i]; heap[i] = t; i = p; }
}
static Ent pop(void) {
    Ent top = heap[0]; heap[0] = heap[--hn]; int64_t i = 0;
    for (;;) { int64_t l = 2 * i + 1, r = l + 1, m = i;
        if (l < hn && heap[l].key > heap[m].key) m = l;
        if (r < hn && heap[r].key > heap[m].key) m = r;
        if (m == i) break; Ent t = heap[m]; heap[m] = heap[i]; heap[i] = t; i = m; }
    return top;
}

int main(int argc, char** argv) {
    if (argc < 4) { fprintf(stderr, "usage: cache_sim stream.bin universe capacity\n"); return 2; }
    FILE* f = fopen(argv[1], "rb"); if (!f) { perror("open"); return 1; }
    fseek(f, 0, SEEK_END); int64_t n = ftell(f) / 4; fseek(f, 0, SEEK_SET);
    int32_t* s = malloc(4 * (n + 1)); if (fread(s, 4, n, f) != (size_t)n) { perror("read"); return 1; } fclose(f);
    const int64_t U = atoll(argv[2]), cap = atoll(argv[3]);
    /* ---- LRU: doubly linked list over resident units ---- */
    int32_t* prev = malloc(4 * U), *next = malloc(4 * U); char* in = calloc(U, 1);
    int32_t head = -1, tail = -1; int64_t resident = 0, lru_miss = 0, distinct = 0;
    char* seen = calloc(U, 1);
    for (int64_t i = 0; i < n; ++i) {
        const int32_t u = s[i];
        if (!seen[u]) { seen[u] = 1; ++distinct; }
        if (in[u]) {  /* unlink */
            if (prev[u] >= 0) next[prev[u]] = next[u]; else head = next[u];
            if (next[u] >= 0) prev[next[u]] = prev[u]; else tail = prev[u];
        } else {
            ++lru_miss;
            if (resident == cap) { const int32_t v = tail; tail = prev[v]; if (tail >= 0) next[tail] = -1; else head = -1; in[v] = 0; --resident; }
            in[u] = 1; ++resident;
        }
        prev[u] = -1; next[u] = head; if (head >= 0) prev[head] = u; head = u; if (tail < 0) tail = u;
    }
    /* ---- Belady: next use of every reference, evict the resident unit whose next use is farthest ---- */
    int64_t* nu = malloc(8 * (n + 1)); int64_t* last = malloc(8 * U);
    for (int64_t u = 0; u < U; ++u) last[u] = INT64_MAX;
    for (int64_t i = n - 1; i >= 0; --i) { nu[i] = last[s[i]]; last[s[i]] = i; }
    int64_t* cur_next = malloc(8 * U);  /* next use of a RESIDENT unit as of now (stale heap entries are skipped) */
    for (int64_t u = 0; u < U; ++u) in[u] = 0;
    heap = malloc(sizeof(Ent) * (n + 8)); hn = 0; resident = 0; int64_t opt_miss = 0;
    for (int64_t i = 0; i < n; ++i) {
        const int32_t u = s[i];
        if (!in[u]) {
            ++opt_miss;
            if (nu[i] == INT64_MAX) continue;  /* never used again: bypass (an optimal policy would not keep it) */
            if (resident == cap) {
                for (;;) { Ent e = pop(); if (in[e.id] && cur_next[e.id] == e.key) { in[e.id] = 0; --resident; break; } }
            }
            in[u] = 1; ++resident;
        } else if (nu[i] == INT64_MAX) { in[u] = 0; --resident; continue; }  /* last use: its place is free */
        cur_next[u] = nu[i]; push(nu[i], u);
    }
    printf("%lld %lld %lld %lld\n", (long long)n, (long long)lru_miss, (long long)opt_miss, (long long)distinct);
    return 0;
}
