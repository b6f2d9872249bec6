// tile_geom.h -- index logic of the tile-fused aggregation schedule (sgm_tile.hip), plain C++ so that the very
// same functions run inside the kernels and inside the CPU model that tests/test_tile_model.py checks against the
// oracle (tests/native/tile_model.cpp).
//
// Schedule.  The 8 (or 5) path recurrences of SURVEY.md Appendix A.4 are chain-sequential, and S = sat16(sum_r L_r)
// needs all of them at every pixel.  Instead of passing a partial S through HBM between chain families, every path
// is first SWEPT over the image keeping only the state with which it ENTERS each T x T tile ("edge states").  A tile
// is then self-contained: from its cost vectors and the entry states of all paths the whole S of the tile is
// rebuilt in LDS and consumed by the winner-take-all without ever being written to HBM.
//
// Paths are addressed as (family, direction):
//   family 0 rows   forward travel (+1, 0)   = path 0 of Appendix A.4, backward = path 4
//   family 1 cols   forward travel ( 0,+1)   = path 2,                 backward = the up-going vertical path
//   family 2 diag   forward travel (+1,+1)   = path 1,                 backward = predecessor (x+1,y+1)
//   family 3 anti   forward travel (-1,+1)   = path 3,                 backward = predecessor (x-1,y+1)
// MODE_SGBM (5 paths) uses both row directions and the three forward (down-going) paths only.
#pragma once

#if defined(__HIPCC__) || defined(__CUDACC__)
#define WASS_HD __host__ __device__ __forceinline__
#else
#define WASS_HD inline
#endif

namespace wass {

enum : int { FAM_ROWS = 0, FAM_COLS = 1, FAM_DIAG = 2, FAM_ANTI = 3 };

// forward direction of travel of a family; the backward path travels (-dx, -dy)
WASS_HD void family_dir(int fam, int& dx, int& dy)
{
    dx = fam == FAM_ROWS ? 1 : (fam == FAM_COLS ? 0 : (fam == FAM_DIAG ? 1 : -1));
    dy = fam == FAM_ROWS ? 0 : 1;
}

// Edge-state arrays of one path: "row edges" hold the state entering a tile through its top/bottom side, indexed
// [tile row][x]; "column edges" the state entering through its left/right side, indexed [tile column][y].
// A pixel entered diagonally through a corner counts as a row-edge entry.
//
// A path travelling (dx,dy) has just finished pixel (x,y): does the next pixel start a new tile, and which slot
// receives the state?  Returns 0 = no store, 1 = row edge, 2 = column edge.
WASS_HD int edge_store_slot(int x, int y, int dx, int dy, int T, int W, int H, long long& idx)
{
    const int nx = x + dx, ny = y + dy;
    if (nx < 0 || nx >= W || ny < 0 || ny >= H) return 0;
    if (dy != 0 && ny / T != y / T) { idx = (long long)(ny / T) * W + nx; return 1; }
    if (dx != 0 && nx / T != x / T) { idx = (long long)(nx / T) * H + ny; return 2; }
    return 0;
}

// The path travelling (dx,dy) enters a tile at pixel (gx,gy): where is its state?  Returns 0 = the predecessor lies
// outside the image (the state is the all-zero border state of Appendix A.4), 1 = row edge, 2 = column edge,
// -1 = (gx,gy) is not an entry pixel (its predecessor is in the same tile).
WASS_HD int edge_entry_slot(int gx, int gy, int dx, int dy, int T, int W, int H, long long& idx)
{
    const int px = gx - dx, py = gy - dy;
    if (px < 0 || px >= W || py < 0 || py >= H) return 0;
    if (dy != 0 && py / T != gy / T) { idx = (long long)(gy / T) * W + gx; return 1; }
    if (dx != 0 && px / T != gx / T) { idx = (long long)(gx / T) * H + gy; return 2; }
    return -1;
}

// One maximal run of a chain inside a tile: n cells starting at local (sx,sy), in the family's forward direction.
struct TileSeg { int sx, sy, n; };

// clip the cells (sx + i dx, sy + i dy), i < n0, to the tw x th cells that exist in a (possibly partial) tile
WASS_HD TileSeg clip_seg(int sx, int sy, int dx, int dy, int n0, int tw, int th)
{
    int i0 = -1, cnt = 0;
    for (int i = 0; i < n0; ++i) {
        const int x = sx + i * dx, y = sy + i * dy;
        if (x >= 0 && x < tw && y >= 0 && y < th) { if (i0 < 0) i0 = i; ++cnt; }
    }
    TileSeg s;
    s.n = cnt;
    s.sx = sx + (i0 < 0 ? 0 : i0) * dx;
    s.sy = sy + (i0 < 0 ? 0 : i0) * dy;
    return s;
}

// Work of wave w (0 <= w < T) of a tile in the phase of family `fam`: at most two runs with at most T cells in all.
//   rows: tile row w.                       cols: tile column w.
//   diag: the diagonals x - y = w - (T-1)  (w + 1 cells) and x - y = w + 1  (T - 1 - w cells).
//   anti: the anti-diagonals x + y = w     (w + 1 cells) and x + y = w + T  (T - 1 - w cells).
// Every cell of the tile belongs to exactly one run of every family.
WASS_HD void tile_segments(int fam, int w, int T, int tw, int th, TileSeg& a, TileSeg& b)
{
    b.sx = b.sy = b.n = 0;
    if (fam == FAM_ROWS) a = clip_seg(0, w, 1, 0, T, tw, th);
    else if (fam == FAM_COLS) a = clip_seg(w, 0, 0, 1, T, tw, th);
    else if (fam == FAM_DIAG) {
        a = clip_seg(0, T - 1 - w, 1, 1, w + 1, tw, th);
        if (w < T - 1) b = clip_seg(w + 1, 0, 1, 1, T - 1 - w, tw, th);
    } else {
        a = clip_seg(w, 0, -1, 1, w + 1, tw, th);
        if (w < T - 1) b = clip_seg(T - 1, w + 1, -1, 1, T - 1 - w, tw, th);
    }
}

// Sizes (in disparity vectors) of the two edge arrays of one path
WASS_HD long long row_edge_vecs(int T, int W, int H) { return (long long)((H + T - 1) / T) * W; }
WASS_HD long long col_edge_vecs(int T, int W, int H) { return (long long)((W + T - 1) / T) * H; }

}  // namespace wass
