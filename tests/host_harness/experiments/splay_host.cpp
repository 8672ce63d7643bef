// splay_host.cpp -- EXPERIMENT (host only, not product, not part of the test suite): how much tree work the GPU
// triangulation would save if a tetrahedron were certified once instead of four times.
//
//   phase A  every point builds its star from its seeds only (no tree);
//   phase B  rounds of "star splaying" (Shewchuk 2005) until the stars agree: a tetrahedron (i,a,b,c) of star i that
//            star a does not have is either unknown to a (i, b, c go to a's inbox) or killed by a vertex of a's link
//            (those of a's link vertices that lie inside the tetrahedron's sphere are inserted into star i);
//   phase C  every tetrahedron is certified against the AABB tree ONCE, in the star of its lowest vertex; a point
//            found goes to the inboxes of all four vertices and phase B resumes.
// Output: rounds, insertions, tree nodes per point, and whether the lists equal a full per-star certification.
#include <algorithm>
#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

#define RF_STAR_FN static inline
#define RF_STAR_NOINLINE static __attribute__((noinline))
#define RF_STAR_NOUNROLL
#include "../../../radfoam_amd/csrc/rf_star.hpp"

using namespace rf::star;
using S64 = Star<64, 124>;

static const int kInbox = 48;

struct Inbox {
    std::atomic<int> n{0};
    uint32_t id[kInbox];
};

static int slot_of(const S64 &s, uint32_t g) {
    for (int k = 0; k < S64::kV; ++k)
        if (s.v[k].use && s.v[k].g == g) return k;
    return -1;
}

static bool has_tet(const S64 &s, uint32_t g0, uint32_t g1, uint32_t g2) {   // link triangle with these three vertices
    const int a = slot_of(s, g0), b = slot_of(s, g1), c = slot_of(s, g2);
    if (a < 0 || b < 0 || c < 0) return false;
    for (int t = 0; t < s.nt; ++t) {
        const int x = s.t[t].a, y = s.t[t].b, z = s.t[t].c;
        if ((x == a || y == a || z == a) && (x == b || y == b || z == b) && (x == c || y == c || z == c)) return true;
    }
    return false;
}

static void push(std::vector<Inbox> &in, uint32_t to, uint32_t id, std::atomic<long> &lost) {
    if (to == kInfinity || id == kInfinity || to == id) return;
    Inbox &b = in[to];
    const int k = b.n.fetch_add(1);
    if (k < kInbox) b.id[k] = id; else lost++;
}

extern "C" int splay_host(const float *pts, uint32_t n, const float *tree, uint32_t depth, uint32_t knn, uint32_t *rows,
                          int stride, uint32_t *degree, double *stats) {
    Tree tr{tree, n, depth};
    std::vector<S64> st(n);
    std::vector<Inbox> inbox(n);
    std::vector<uint8_t> dirty(n, 1), failed(n, 0);
    std::atomic<long> inserts{0}, lost{0}, nodes{0}, incons{0}, certs{0}, cert_hits{0};
    // ---- phase A
#pragma omp parallel for schedule(dynamic, 256)
    for (uint32_t i = 0; i < n; ++i) {
        S64 &s = st[i];
        star_reset(s, i, pts + 3 * (size_t)i);
        uint32_t seeds[64];
        int ns = 0;
        const uint32_t b0 = i & ~63u, b1 = b0 + 64 < n ? b0 + 64 : n;
        float d2[64];
        for (uint32_t k = b0; k < b1; ++k) {
            const float dx = pts[3 * k] - pts[3 * i], dy = pts[3 * k + 1] - pts[3 * i + 1], dz = pts[3 * k + 2] - pts[3 * i + 2];
            d2[k - b0] = k == i ? 3.4e38f : dx * dx + dy * dy + dz * dz;
        }
        for (uint32_t r = 0; r < knn; ++r) {
            int best = -1;
            for (uint32_t k = 0; k < b1 - b0; ++k)
                if (d2[k] < 3.4e38f && (best < 0 || d2[k] < d2[best])) best = (int)k;
            if (best < 0) break;
            seeds[ns++] = b0 + (uint32_t)best;
            d2[best] = 3.4e38f;
        }
        uint32_t ins = 0;
        star_seed(s, pts, seeds, ns, ins);
        inserts += ins;
        if (s.status != kOk) failed[i] = 1;
    }
    int rounds = 0, outer = 0;
    for (;; ++outer) {
        // ---- phase B: splay until quiet
        for (;;) {
            ++rounds;
            std::atomic<long> changed{0};
            // 1. inboxes
#pragma omp parallel for schedule(dynamic, 256)
            for (uint32_t i = 0; i < n; ++i) {
                Inbox &b = inbox[i];
                const int m = std::min(b.n.load(), kInbox);
                if (!m || failed[i]) { b.n = 0; continue; }
                S64 &s = st[i];
                for (int k = 0; k < m; ++k) {
                    const uint32_t g = b.id[k];
                    if (slot_of(s, g) >= 0) continue;
                    const int r = star_insert(s, g, pts + 3 * (size_t)g);
                    if (r < 0) { failed[i] = 1; break; }
                    if (r > 0) { inserts++; changed++; dirty[i] = 1; }
                }
                b.n = 0;
            }
            // 2. compare with the neighbours' stars (read-only on the others; own insertions are deferred to a list)
            std::vector<std::vector<uint32_t>> todo(n);
#pragma omp parallel for schedule(dynamic, 256)
            for (uint32_t i = 0; i < n; ++i) {
                if (failed[i]) continue;
                const S64 &s = st[i];
                for (int t = 0; t < s.nt; ++t) {
                    const uint32_t g[3] = {s.v[s.t[t].a].g, s.v[s.t[t].b].g, s.v[s.t[t].c].g};
                    for (int e = 0; e < 3; ++e) {
                        const uint32_t x = g[e];
                        if (x == kInfinity || failed[x]) continue;
                        if (!dirty[i] && !dirty[x]) continue;   // neither changed since they were last compared
                        const uint32_t o1 = g[(e + 1) % 3], o2 = g[(e + 2) % 3];
                        const S64 &sx = st[x];
                        if (has_tet(sx, i, o1, o2)) continue;
                        incons++;
                        // what x does not know yet
                        if (slot_of(sx, i) < 0) push(inbox, x, i, lost);
                        if (o1 != kInfinity && slot_of(sx, o1) < 0) push(inbox, x, o1, lost);
                        if (o2 != kInfinity && slot_of(sx, o2) < 0) push(inbox, x, o2, lost);
                        // what x knows and this tetrahedron cannot live with
                        for (int k = 1; k < S64::kV; ++k) {
                            if (!sx.v[k].use) continue;
                            const uint32_t d = sx.v[k].g;
                            if (d == i || d == g[0] || d == g[1] || d == g[2]) continue;
                            const float q[3] = {sx.v[k].x, sx.v[k].y, sx.v[k].z};
                            if (conflict(s, t, q)) todo[i].push_back(d);
                        }
                    }
                }
            }
#pragma omp parallel for schedule(dynamic, 256)
            for (uint32_t i = 0; i < n; ++i) dirty[i] = 0;
#pragma omp parallel for schedule(dynamic, 256)
            for (uint32_t i = 0; i < n; ++i) {
                if (failed[i] || todo[i].empty()) continue;
                S64 &s = st[i];
                std::sort(todo[i].begin(), todo[i].end());
                todo[i].erase(std::unique(todo[i].begin(), todo[i].end()), todo[i].end());
                for (uint32_t g : todo[i]) {
                    if (slot_of(s, g) >= 0) continue;
                    const int r = star_insert(s, g, pts + 3 * (size_t)g);
                    if (r < 0) { failed[i] = 1; break; }
                    if (r > 0) { inserts++; changed++; dirty[i] = 1; }
                }
            }
            long pending = 0;
            for (uint32_t i = 0; i < n; ++i) pending += inbox[i].n.load() > 0;
            fprintf(stderr, "[splay] outer %d round %d: changed %ld, inconsistencies so far %ld, inboxes pending %ld\n", outer,
                    rounds, changed.load(), incons.load(), pending);
            if (changed == 0 && pending == 0) break;
            if (rounds > 60) break;
        }
        // ---- phase C: each tetrahedron certified once, by its lowest vertex
        std::atomic<long> found{0};
#pragma omp parallel for schedule(dynamic, 256)
        for (uint32_t i = 0; i < n; ++i) {
            if (failed[i]) continue;
            S64 &s = st[i];
            const HullSet all{nullptr, 0, 0xFFFFFFFFu};
            for (int t = 0; t < s.nt; ++t) {
                if (s.t[t].f & kCertified) continue;
                const uint32_t g[3] = {s.v[s.t[t].a].g, s.v[s.t[t].b].g, s.v[s.t[t].c].g};
                if (g[0] < i || g[1] < i || g[2] < i) continue;   // someone lower owns it (infinity is the highest id)
                float q[3];
                uint32_t vis = 0;
                const uint32_t j = star_search(s, tr, pts, t, all, q, vis);
                nodes += vis;
                certs++;
                if (j == kInfinity) { s.t[t].f |= kCertified; continue; }
                cert_hits++;
                found++;
                push(inbox, i, j, lost);
                for (int e = 0; e < 3; ++e) push(inbox, g[e], j, lost);
            }
        }
        fprintf(stderr, "[splay] outer %d: certification found %ld points\n", outer, found.load());
        if (found == 0) break;
        if (outer > 20) break;
    }
    long nfailed = 0;
#pragma omp parallel for schedule(dynamic, 256) reduction(+ : nfailed)
    for (uint32_t i = 0; i < n; ++i) {
        if (failed[i]) { degree[i] = 0; nfailed++; continue; }
        bool h;
        degree[i] = (uint32_t)star_neighbours(st[i], rows + (size_t)i * stride, 1, &h);
    }
    stats[0] = rounds; stats[1] = (double)inserts / n; stats[2] = (double)nodes / n; stats[3] = (double)incons / n;
    stats[4] = (double)certs / n; stats[5] = (double)cert_hits; stats[6] = (double)lost; stats[7] = (double)nfailed;
    return 0;
}
