// Native sphere-mesh producer: Fibonacci points -> spherical Delaunay (3-D convex hull of the
// unit-sphere points, pole included) -> triangles/halfedges -> CSR neighbour graph.
//
// This replaces the reference's input producer for the hot path:
//   generateFibonacciSphere      js/sphere-mesh.js:9-37
//   stereographic Delaunay + addPoleToMesh   js/sphere-mesh.js:41-90,174-186  (delaunator@5.0.1 there)
//   SphereMesh CSR construction  js/sphere-mesh.js:94-146
//   computeNeighborDist          js/sphere-mesh.js:191-203
//
// Design (not a port): the reference triangulates a stereographic projection with a sweep-hull and then
// stitches the projection pole back in.  The planar Delaunay of the projected points plus that pole fan
// is exactly the convex hull of {points} U {pole}; we build that hull directly and in parallel: every
// point computes its own star by gift-wrapping the candidates found in a hashed 3-D grid, with an
// empty-circumcap certificate that widens the search when needed.  All orientation predicates are
// evaluated on the index-sorted 4-tuple so every star sees bit-identical decisions, which makes the
// independently computed stars mutually consistent.  Triangles are wound counter-clockwise seen from
// outside the sphere.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <atomic>
#include <string>
#include <vector>

#include "host_util.h"
#include "wo_internal.h"

namespace wo {

// ---------------------------------------------------------------------------------------------
// Points (js/sphere-mesh.js:9-37).  Sequential because the reference draws four LCG values per
// point in a fixed order; evaluation order of every expression follows the JS left-to-right rules.
// ---------------------------------------------------------------------------------------------
void fib_sphere_points(int N, double jitter, double seed, float* xyz /* 3*(N+1) */) {
    ParkMiller rng(seed);
    const double PI = 3.141592653589793;
    const double s = 3.6 / std::sqrt((double)N);
    const double dlong = PI * (3.0 - std::sqrt(5.0));
    const double dz = 2.0 / (double)N;
    double lng = 0.0, z = 1.0 - dz / 2.0;
    for (int k = 0; k < N; ++k) {
        const double r = std::sqrt(1.0 - z * z);
        double latDeg = std::asin(z) * 180.0 / PI;
        double lonDeg = lng * 180.0 / PI;
        if (jitter > 0) {
            double a = rng.next(); double b = rng.next();
            const double jLat = a - b;
            a = rng.next(); b = rng.next();
            const double jLon = a - b;
            const double nextZ = std::max(-1.0, z - dz * 2.0 * PI * r / s);
            latDeg += jitter * jLat * (latDeg - std::asin(nextZ) * 180.0 / PI);
            lonDeg += jitter * jLon * (s / r * 180.0 / PI);
        }
        const double latR = latDeg * PI / 180.0;
        const double lonR = lonDeg * PI / 180.0;
        xyz[3 * k]     = (float)(std::cos(latR) * std::cos(lonR));
        xyz[3 * k + 1] = (float)(std::cos(latR) * std::sin(lonR));
        xyz[3 * k + 2] = (float)(std::sin(latR));
        lng += dlong;
        z -= dz;
    }
    // pole region appended by buildSphere (js/sphere-mesh.js:179-181)
    xyz[3 * N] = 0.f; xyz[3 * N + 1] = 0.f; xyz[3 * N + 2] = 1.f;
}

// ---------------------------------------------------------------------------------------------
// Spherical Delaunay
// ---------------------------------------------------------------------------------------------
namespace {

struct P3 { double x, y, z; };

// det[b-a, c-a, d-a] evaluated on the index-sorted tuple; sign fixed up by permutation parity so the
// floating-point value (and therefore every decision) is identical from whichever star asks.
struct Orient {
    const P3* pts;
    inline double raw(int a, int b, int c, int d) const {
        const P3 &A = pts[a], &B = pts[b], &C = pts[c], &D = pts[d];
        const double bx = B.x - A.x, by = B.y - A.y, bz = B.z - A.z;
        const double cx = C.x - A.x, cy = C.y - A.y, cz = C.z - A.z;
        const double dx = D.x - A.x, dy = D.y - A.y, dz = D.z - A.z;
        return bx * (cy * dz - cz * dy) - by * (cx * dz - cz * dx) + bz * (cx * dy - cy * dx);
    }
    inline double operator()(int a, int b, int c, int d) const {
        int v[4] = {a, b, c, d};
        int sgn = 1;
        // 4-element sorting network, tracking parity
#define WO_CSWAP(i, j) if (v[i] > v[j]) { int t = v[i]; v[i] = v[j]; v[j] = t; sgn = -sgn; }
        WO_CSWAP(0, 1) WO_CSWAP(2, 3) WO_CSWAP(0, 2) WO_CSWAP(1, 3) WO_CSWAP(1, 2)
#undef WO_CSWAP
        return sgn * raw(v[0], v[1], v[2], v[3]);
    }
};

struct Grid {
    double h;       // cell edge
    int G;          // cells per axis
    std::vector<uint64_t> cellKey;   // sorted unique keys
    std::vector<int> cellStart;      // start into sortedIdx, size = cells+1
    std::vector<int> sortedIdx;      // point ids grouped by cell
    std::vector<uint64_t> hkey;      // open-addressing table: key+1 (0 = empty)
    std::vector<int> hval;
    uint64_t hmask;

    inline int coord(double v) const {
        int c = (int)std::floor((v + 1.0) / h);
        if (c < 0) c = 0;
        if (c >= G) c = G - 1;
        return c;
    }
    inline uint64_t key(int ix, int iy, int iz) const {
        return (uint64_t)ix + (uint64_t)G * ((uint64_t)iy + (uint64_t)G * (uint64_t)iz);
    }
    static inline uint64_t mix(uint64_t k) {
        k ^= k >> 33; k *= 0xff51afd7ed558ccdULL; k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ULL; k ^= k >> 33;
        return k;
    }
    inline int find(uint64_t k) const {
        uint64_t slot = mix(k) & hmask;
        for (;;) {
            uint64_t v = hkey[slot];
            if (v == 0) return -1;
            if (v == k + 1) return hval[slot];
            slot = (slot + 1) & hmask;
        }
    }
};

void radix_sort_pairs(std::vector<uint64_t>& keys, std::vector<int>& vals, int bits) {
    const size_t n = keys.size();
    std::vector<uint64_t> k2(n);
    std::vector<int> v2(n);
    for (int shift = 0; shift < bits; shift += 11) {
        size_t cnt[2049];
        std::memset(cnt, 0, sizeof(cnt));
        for (size_t i = 0; i < n; ++i) cnt[((keys[i] >> shift) & 2047) + 1]++;
        for (int b = 0; b < 2048; ++b) cnt[b + 1] += cnt[b];
        for (size_t i = 0; i < n; ++i) {
            size_t d = cnt[(keys[i] >> shift) & 2047]++;
            k2[d] = keys[i]; v2[d] = vals[i];
        }
        keys.swap(k2); vals.swap(v2);
    }
}

void build_grid(Grid& g, const P3* pts, int V) {
    const double spacing = std::sqrt(4.0 * 3.141592653589793 / (double)V);
    g.h = 2.0 * spacing;
    if (g.h > 0.5) g.h = 0.5;
    g.G = (int)std::ceil(2.0 / g.h) + 1;
    std::vector<uint64_t> keys(V);
    std::vector<int> vals(V);
    parallel_ranges(V, [&](int64_t b, int64_t e, int) {
        for (int64_t i = b; i < e; ++i) {
            keys[i] = g.key(g.coord(pts[i].x), g.coord(pts[i].y), g.coord(pts[i].z));
            vals[i] = (int)i;
        }
    });
    int bits = 1;
    { uint64_t mx = (uint64_t)g.G * g.G * g.G; while ((1ULL << bits) < mx) ++bits; }
    radix_sort_pairs(keys, vals, bits);
    g.sortedIdx.swap(vals);
    g.cellKey.clear(); g.cellStart.clear();
    for (int i = 0; i < V; ++i) {
        if (i == 0 || keys[i] != keys[i - 1]) { g.cellKey.push_back(keys[i]); g.cellStart.push_back(i); }
    }
    g.cellStart.push_back(V);
    size_t cap = 16;
    while (cap < g.cellKey.size() * 2 + 8) cap <<= 1;
    g.hkey.assign(cap, 0); g.hval.assign(cap, -1); g.hmask = cap - 1;
    for (size_t c = 0; c < g.cellKey.size(); ++c) {
        uint64_t slot = Grid::mix(g.cellKey[c]) & g.hmask;
        while (g.hkey[slot] != 0) slot = (slot + 1) & g.hmask;
        g.hkey[slot] = g.cellKey[c] + 1; g.hval[slot] = (int)c;
    }
}

constexpr int MAX_STAR = 48;

// Star of p (neighbours counter-clockwise seen from outside).  Returns degree, or <0 on failure.
int star_of(int p, const P3* pts, int V, const Grid& g, const Orient& orient, int* out,
            std::vector<int>& cand) {
    const P3 P = pts[p];
    const int cx = g.coord(P.x), cy = g.coord(P.y), cz = g.coord(P.z);
    int failcode = -1;
    for (int level = 1; level <= 6; ++level) {
        cand.clear();
        const bool brute = (level == 6) || ((2 * level + 1) >= g.G);
        if (brute) {
            for (int i = 0; i < V; ++i) if (i != p) cand.push_back(i);
        } else {
            for (int dz = -level; dz <= level; ++dz) {
                int iz = cz + dz; if (iz < 0 || iz >= g.G) continue;
                for (int dy = -level; dy <= level; ++dy) {
                    int iy = cy + dy; if (iy < 0 || iy >= g.G) continue;
                    for (int dx = -level; dx <= level; ++dx) {
                        int ix = cx + dx; if (ix < 0 || ix >= g.G) continue;
                        int c = g.find(g.key(ix, iy, iz));
                        if (c < 0) continue;
                        for (int k = g.cellStart[c]; k < g.cellStart[c + 1]; ++k) {
                            int q = g.sortedIdx[k];
                            if (q != p) cand.push_back(q);
                        }
                    }
                }
            }
        }
        if (cand.size() < 3) continue;
        // nearest candidate (lowest index on ties) is always a Delaunay neighbour
        int q0 = -1; double best = 1e300;
        for (int q : cand) {
            const double dx = pts[q].x - P.x, dy = pts[q].y - P.y, dz = pts[q].z - P.z;
            const double d2 = dx * dx + dy * dy + dz * dz;
            if (d2 < best || (d2 == best && q < q0)) { best = d2; q0 = q; }
        }
        int deg = 0; bool ok = true; bool certified = true;
        // searched block in coordinates (cells cx-level .. cx+level); the grid origin is -1
        const double lox = (cx - level) * g.h - 1.0, hix = (cx + level + 1) * g.h - 1.0;
        const double loy = (cy - level) * g.h - 1.0, hiy = (cy + level + 1) * g.h - 1.0;
        const double loz = (cz - level) * g.h - 1.0, hiz = (cz + level + 1) * g.h - 1.0;
        int q = q0;
        for (;;) {
            if (deg >= MAX_STAR) { ok = false; failcode = -2;
                break; }
            out[deg++] = q;
            // Jarvis step: r such that every other candidate lies on the origin side of plane (p,q,r)
            int r = -1;
            for (int s : cand) {
                if (s == q) continue;
                if (r < 0) { r = s; continue; }
                if (orient(p, q, r, s) > 0.0) r = s;
            }
            // circumcap of (p,q,r): centre direction n = (q-p)x(r-p) (outward).  Every point of the cap is
            // within chord distance |c - p| of the centre c = n/|n|; the face is certified empty when that
            // ball lies inside the searched block of grid cells.
            {
                const P3 &Q = pts[q], &R = pts[r];
                const double ax = Q.x - P.x, ay = Q.y - P.y, az = Q.z - P.z;
                const double bx = R.x - P.x, by = R.y - P.y, bz = R.z - P.z;
                double nx = ay * bz - az * by, ny = az * bx - ax * bz, nz = ax * by - ay * bx;
                const double nl = std::sqrt(nx * nx + ny * ny + nz * nz);
                if (!(nl > 0.0)) { ok = false; failcode = -3; break; }
                nx /= nl; ny /= nl; nz /= nl;
                if (nx * P.x + ny * P.y + nz * P.z <= 0.0) { certified = false; }
                else {
                    const double ex = nx - P.x, ey = ny - P.y, ez = nz - P.z;
                    const double rad = std::sqrt(ex * ex + ey * ey + ez * ez) * 1.000001 + 1e-12;
                    if (nx - rad < lox || nx + rad > hix || ny - rad < loy || ny + rad > hiy ||
                        nz - rad < loz || nz + rad > hiz) certified = false;
                }
            }
            q = r;
            if (q == q0) break;
        }
        if (!ok) continue;
        if (brute || certified) return deg;
    }
    return failcode;
}

}  // namespace

// triangles/halfedges sized 3*(2V-4).  Returns 0 on success.
int sphere_delaunay(int V, const float* xyz, int* triangles, int* halfedges, std::string& err) {
    if (V < 4) { err = "sphere_delaunay: need at least 4 points"; return 1; }
    std::vector<P3> pts(V);
    // The float32 inputs sit up to ~3e-8 off the unit sphere, enough to push one of two nearly
    // coincident points (they occur from ~1e6 jittered points up) inside the hull of the others; a planar
    // Delaunay of the projection keeps every point, so we take the radial noise out before building the hull.
    for (int i = 0; i < V; ++i) {
        const double x = xyz[3 * i], y = xyz[3 * i + 1], z = xyz[3 * i + 2];
        const double l = std::sqrt(x * x + y * y + z * z);
        if (!(l > 0.0)) { err = "sphere_delaunay: zero-length point"; return 1; }
        pts[i] = {x / l, y / l, z / l};
    }
    Grid g;
    build_grid(g, pts.data(), V);
    Orient orient{pts.data()};

    // pass 1: stars (degree + neighbours, rotated to start at the lowest-index neighbour)
    std::vector<int> deg(V);
    std::vector<int> starBuf((size_t)V * 12);         // common case storage
    std::vector<std::vector<int>> bigStar;            // rare degree > 12
    std::vector<int> bigIndex(V, -1);
    std::atomic<int> failed{0};
    std::vector<std::vector<std::pair<int, std::vector<int>>>> bigLocal(host_threads() + 1);
    parallel_ranges(V, [&](int64_t b, int64_t e, int tid) {
        std::vector<int> cand; cand.reserve(256);
        int st[MAX_STAR];
        for (int64_t p = b; p < e; ++p) {
            int d = star_of((int)p, pts.data(), V, g, orient, st, cand);
            if (d < 3) { if (failed.fetch_add(1) == 0) fprintf(stderr, "star fail p=%d code=%d xyz=%.9g %.9g %.9g\n", (int)p, d, pts[p].x, pts[p].y, pts[p].z); deg[p] = 0; continue; }
            int m = 0;
            for (int i = 1; i < d; ++i) if (st[i] < st[m]) m = i;
            deg[p] = d;
            if (d <= 12) {
                for (int i = 0; i < d; ++i) starBuf[(size_t)p * 12 + i] = st[(m + i) % d];
            } else {
                std::vector<int> v(d);
                for (int i = 0; i < d; ++i) v[i] = st[(m + i) % d];
                bigLocal[tid].emplace_back((int)p, std::move(v));
            }
        }
    }, 1024);
    if (failed.load() != 0) { err = "sphere_delaunay: star construction failed for " + std::to_string(failed.load()) + " points"; return 2; }
    for (auto& l : bigLocal) for (auto& pr : l) { bigIndex[pr.first] = (int)bigStar.size(); bigStar.push_back(std::move(pr.second)); }
    auto star = [&](int p) -> const int* { return bigIndex[p] >= 0 ? bigStar[bigIndex[p]].data() : &starBuf[(size_t)p * 12]; };

    // pass 2: triangle ownership (owner = lowest vertex), ids by (owner, star position)
    std::vector<int64_t> starOff(V + 1, 0);
    for (int p = 0; p < V; ++p) starOff[p + 1] = starOff[p] + deg[p];
    std::vector<int> triOfStar(starOff[V], -1);
    std::vector<int> ownCount(V + 1, 0);
    parallel_ranges(V, [&](int64_t b, int64_t e, int) {
        for (int64_t p = b; p < e; ++p) {
            const int* s = star((int)p); int d = deg[p], c = 0;
            for (int i = 0; i < d; ++i) { int q = s[i], r = s[(i + 1) % d]; if (p < q && p < r) ++c; }
            ownCount[p + 1] = c;
        }
    });
    for (int p = 0; p < V; ++p) ownCount[p + 1] += ownCount[p];
    const int T = ownCount[V];
    if (T != 2 * V - 4) { err = "sphere_delaunay: triangle count " + std::to_string(T) + " != 2V-4 (inconsistent stars)"; return 3; }
    parallel_ranges(V, [&](int64_t b, int64_t e, int) {
        for (int64_t p = b; p < e; ++p) {
            const int* s = star((int)p); int d = deg[p]; int t = ownCount[p];
            for (int i = 0; i < d; ++i) {
                int q = s[i], r = s[(i + 1) % d];
                if (p < q && p < r) {
                    triangles[3 * t] = (int)p; triangles[3 * t + 1] = q; triangles[3 * t + 2] = r;
                    triOfStar[starOff[p] + i] = t; ++t;
                }
            }
        }
    });
    // pass 3: halfedges.  Side s = (u -> v) of triangle t; its twin is side (v -> u) of the triangle
    // (v, u, w) where w follows u in v's star.
    std::atomic<int> bad{0};
    parallel_ranges(T, [&](int64_t b, int64_t e, int) {
        for (int64_t t = b; t < e; ++t) {
            for (int k = 0; k < 3; ++k) {
                const int u = triangles[3 * t + k], v = triangles[3 * t + (k + 1) % 3];
                const int* sv = star(v); const int dv = deg[v];
                int i = -1;
                for (int j = 0; j < dv; ++j) if (sv[j] == u) { i = j; break; }
                if (i < 0) { bad.fetch_add(1); halfedges[3 * t + k] = -1; continue; }
                const int w = sv[(i + 1) % dv];
                // owner of (v,u,w) and the star position of that triangle in the owner's star
                int m = v, a = u;                       // triangle as (m, a, .) counter-clockwise
                if (u < m && u < w) { m = u; a = w; }
                else if (w < m && w < u) { m = w; a = v; }
                const int* sm = star(m); const int dm = deg[m];
                int j2 = -1;
                for (int j = 0; j < dm; ++j) if (sm[j] == a) { j2 = j; break; }
                int t2 = (j2 >= 0) ? triOfStar[starOff[m] + j2] : -1;
                if (t2 < 0) { bad.fetch_add(1); halfedges[3 * t + k] = -1; continue; }
                int side = -1;
                for (int c = 0; c < 3; ++c) if (triangles[3 * t2 + c] == v && triangles[3 * t2 + (c + 1) % 3] == u) side = c;
                if (side < 0) { bad.fetch_add(1); halfedges[3 * t + k] = -1; continue; }
                halfedges[3 * t + k] = 3 * t2 + side;
            }
        }
    });
    if (bad.load() != 0) { err = "sphere_delaunay: " + std::to_string(bad.load()) + " unmatched half-edges (inconsistent stars)"; return 4; }
    return 0;
}

// ---------------------------------------------------------------------------------------------
// CSR from triangles/halfedges, following the circulation rule of js/sphere-mesh.js:102-141:
// row r starts at the lowest-numbered side leaving r and steps s <- next(halfedges[s]).
// ---------------------------------------------------------------------------------------------
static inline int next_side(int s) { return (s % 3 == 2) ? s - 2 : s + 1; }

int mesh_csr(int V, int numSides, const int* triangles, const int* halfedges,
             int* adjOffset /*V+1*/, int* adjList /*numSides*/, int* adjTri /*numSides or null*/, std::string& err) {
    if (V < 0 || numSides < 0 || numSides % 3 != 0) { err = "mesh_csr: bad sizes"; return 3; }
    {   // the walks below index r_s[] with triangle corners and follow half-edges: refuse anything out of range first
        std::atomic<int> oob{0};
        parallel_ranges(numSides, [&](int64_t b, int64_t e, int) {
            for (int64_t s = b; s < e; ++s) {
                if (triangles[s] < 0 || triangles[s] >= V) { oob = 1; break; }
                if (halfedges[s] < -1 || halfedges[s] >= numSides) { oob = 2; break; }
            }
        });
        if (oob == 1) { err = "mesh_csr: triangle corner out of range"; return 3; }
        if (oob == 2) { err = "mesh_csr: half-edge index out of range"; return 3; }
    }
    std::vector<int> r_s(V, -1);
    for (int s = numSides - 1; s >= 0; --s) r_s[triangles[s]] = s;   // lowest side wins
    adjOffset[0] = 0;
    std::vector<int> cnt(V, 0);
    std::atomic<int> bad{0};
    parallel_ranges(V, [&](int64_t b, int64_t e, int) {
        for (int64_t r = b; r < e; ++r) {
            int s0 = r_s[r]; if (s0 < 0) continue;
            int s = s0, c = 0;
            do { ++c; int h = halfedges[s]; if (h < 0 || c > 4096) { bad.fetch_add(1); break; } s = next_side(h); } while (s != s0);
            cnt[r] = c;
        }
    });
    if (bad.load()) { err = "mesh_csr: open or malformed half-edge structure"; return 1; }
    for (int r = 0; r < V; ++r) adjOffset[r + 1] = adjOffset[r] + cnt[r];
    if (adjOffset[V] > numSides) { err = "mesh_csr: adjacency larger than side count"; return 2; }
    parallel_ranges(V, [&](int64_t b, int64_t e, int) {
        for (int64_t r = b; r < e; ++r) {
            int s0 = r_s[r]; if (s0 < 0) continue;
            int s = s0, idx = adjOffset[r];
            do {
                adjList[idx] = triangles[next_side(s)];
                if (adjTri) adjTri[idx] = s / 3;
                ++idx;
                s = next_side(halfedges[s]);
            } while (s != s0);
        }
    });
    return 0;
}

void neighbor_dist(int V, const int* adjOffset, const int* adjList, const float* xyz, float* out) {
    parallel_ranges(V, [&](int64_t b, int64_t e, int) {
        for (int64_t r = b; r < e; ++r) {
            const double x = xyz[3 * r], y = xyz[3 * r + 1], z = xyz[3 * r + 2];
            for (int i = adjOffset[r]; i < adjOffset[r + 1]; ++i) {
                const int nb = adjList[i];
                const double dx = x - (double)xyz[3 * nb], dy = y - (double)xyz[3 * nb + 1], dz = z - (double)xyz[3 * nb + 2];
                out[i] = (float)std::sqrt(dx * dx + dy * dy + dz * dz);
            }
        }
    });
}

}  // namespace wo
