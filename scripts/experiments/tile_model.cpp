// tile_model.cpp -- scalar CPU model of the tile-fused aggregation schedule (wass_amd/csrc/sgm_tile.hip), built on
// the SAME index functions the kernels use (wass_amd/csrc/tile_geom.h).  Test infrastructure: tests/test_tile_model.py
// compares its S volume with the oracle's, which pins the edge-state layout, the entry/store conventions and the
// per-wave work split of a tile before any GPU is involved.
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "tile_geom.h"

using namespace wass;

namespace {

// one step of Appendix A.4 on a normalised predecessor state N (min_d N = 0):  L(d) = C(d) + min(N(d), N(d-1)+P1,
// N(d+1)+P1, P2);  the state handed on is L - min_d L
void step(const int16_t* C, int D, int P1, int P2, std::vector<int>& N, std::vector<int>& L)
{
    int mn = 1 << 30;
    for (int d = 0; d < D; ++d) {
        int v = N[d];
        if (d > 0 && N[d - 1] + P1 < v) v = N[d - 1] + P1;
        if (d + 1 < D && N[d + 1] + P1 < v) v = N[d + 1] + P1;
        if (P2 < v) v = P2;
        L[d] = C[d] + v;
        if (L[d] < mn) mn = L[d];
    }
    for (int d = 0; d < D; ++d) N[d] = L[d] - mn;
}

struct Edges {
    std::vector<int> row, col;   // [slots][D]
    std::vector<char> row_set, col_set;
};

}  // namespace

extern "C" int tile_model_S(const int16_t* C, int W, int H, int D, int P1, int P2, int T, int ndirs, int16_t* S_out,
                            int* cover_errors)
{
    const long long nrow = row_edge_vecs(T, W, H), ncol = col_edge_vecs(T, W, H);
    Edges E[4][2];
    // ---- phase 1: one sweep per path, keeping only the states that enter a tile
    for (int fam = 0; fam < 4; ++fam)
        for (int dir = 0; dir < 2; ++dir) {
            if (dir == 1 && ndirs == 5 && fam != FAM_ROWS) continue;
            int dx, dy;
            family_dir(fam, dx, dy);
            if (dir) { dx = -dx; dy = -dy; }
            Edges& e = E[fam][dir];
            e.row.assign((size_t)nrow * D, -1); e.col.assign((size_t)ncol * D, -1);
            e.row_set.assign(nrow, 0); e.col_set.assign(ncol, 0);
            std::vector<int> N(D), L(D);
            for (int y0 = 0; y0 < H; ++y0)
                for (int x0 = 0; x0 < W; ++x0) {
                    const int px = x0 - dx, py = y0 - dy;
                    if (px >= 0 && px < W && py >= 0 && py < H) continue;       // not the head of a chain
                    std::fill(N.begin(), N.end(), 0);
                    for (int x = x0, y = y0; x >= 0 && x < W && y >= 0 && y < H; x += dx, y += dy) {
                        step(C + ((size_t)y * W + x) * D, D, P1, P2, N, L);
                        long long idx = 0;
                        const int k = edge_store_slot(x, y, dx, dy, T, W, H, idx);
                        if (k == 1) { if (idx < 0 || idx >= nrow) return -2; memcpy(&e.row[(size_t)idx * D], N.data(), D * sizeof(int)); e.row_set[idx] = 1; }
                        if (k == 2) { if (idx < 0 || idx >= ncol) return -2; memcpy(&e.col[(size_t)idx * D], N.data(), D * sizeof(int)); e.col_set[idx] = 1; }
                    }
                }
        }
    // ---- phase 2: every tile on its own
    std::vector<int> S((size_t)W * H * D, 0);
    std::vector<int> cover((size_t)W * H * 8, 0);
    int errors = 0;
    const int ntx = (W + T - 1) / T, nty = (H + T - 1) / T;
    std::vector<int> N(D), L(D);
    for (int ty = 0; ty < nty; ++ty)
        for (int tx = 0; tx < ntx; ++tx) {
            const int X0 = tx * T, Y0 = ty * T;
            const int tw = W - X0 < T ? W - X0 : T, th = H - Y0 < T ? H - Y0 : T;
            for (int fam = 0; fam < 4; ++fam)
                for (int w = 0; w < T; ++w) {
                    TileSeg sg[2];
                    tile_segments(fam, w, T, tw, th, sg[0], sg[1]);
                    if (sg[0].n + sg[1].n > T) return -3;
                    int fdx, fdy;
                    family_dir(fam, fdx, fdy);
                    for (int q = 0; q < 2; ++q) {
                        if (sg[q].n == 0) continue;
                        for (int dir = 0; dir < 2; ++dir) {
                            if (dir == 1 && ndirs == 5 && fam != FAM_ROWS) continue;
                            const int dx = dir ? -fdx : fdx, dy = dir ? -fdy : fdy;
                            // entry cell: first cell of the run for the forward path, last one for the backward path
                            const int ex = X0 + sg[q].sx + (dir ? (sg[q].n - 1) * fdx : 0);
                            const int ey = Y0 + sg[q].sy + (dir ? (sg[q].n - 1) * fdy : 0);
                            long long idx = 0;
                            const int k = edge_entry_slot(ex, ey, dx, dy, T, W, H, idx);
                            const Edges& e = E[fam][dir];
                            if (k == 0) std::fill(N.begin(), N.end(), 0);
                            else if (k == 1) { if (!e.row_set[idx]) ++errors; for (int d = 0; d < D; ++d) N[d] = e.row[(size_t)idx * D + d]; }
                            else if (k == 2) { if (!e.col_set[idx]) ++errors; for (int d = 0; d < D; ++d) N[d] = e.col[(size_t)idx * D + d]; }
                            else { ++errors; std::fill(N.begin(), N.end(), 0); }
                            for (int i = 0, x = ex, y = ey; i < sg[q].n; ++i, x += dx, y += dy) {
                                if (x < X0 || x >= X0 + tw || y < Y0 || y >= Y0 + th) return -4;
                                const size_t p = (size_t)y * W + x;
                                step(C + p * D, D, P1, P2, N, L);
                                for (int d = 0; d < D; ++d) S[p * D + d] += L[d];
                                cover[p * 8 + fam * 2 + dir]++;
                            }
                        }
                    }
                }
        }
    for (size_t p = 0; p < (size_t)W * H; ++p)
        for (int f = 0; f < 4; ++f)
            for (int dir = 0; dir < 2; ++dir) {
                const int want = (dir == 1 && ndirs == 5 && f != FAM_ROWS) ? 0 : 1;
                if (cover[p * 8 + f * 2 + dir] != want) ++errors;
            }
    for (size_t i = 0; i < S.size(); ++i) S_out[i] = (int16_t)(S[i] > 32767 ? 32767 : S[i]);
    if (cover_errors) *cover_errors = errors;
    return 0;
}
