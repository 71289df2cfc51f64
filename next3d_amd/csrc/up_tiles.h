// Tile plan of the pre-split transposed 3x3 stride-2 kernels (conv2d_ps_bf16x3.hip: conv2d_up_ps_body, conv2d_f16.hip:
// conv2d_up_f16_body).  Those kernels walk the (H+1) x (W+1) POSITION grid of the layer (position (gy, gx) owns the outputs
// (2 gy + a, 2 gx + b) and reads the inputs (gy - 1 .. gy, gx - 1 .. gx)); a workgroup takes th x tw <= 256 positions whose
// (th+1) x (tw+1) patch fits 297 slots.  H and W are powers of two, so tiling the whole (H+1) x (W+1) grid with 8 x 32 tiles
// wastes a column of tiles for ONE position column (129 = 4 x 32 + 1: 85 tiles where 66 cover the area — 22 % of the MFMA work
// of the 128x128 layers, 13 % at 256, 10 % at 64).  Plan: the H x W part in exact 8 x 32 tiles, the last position row (gy = H,
// only its even output rows exist) and the last position column (gx = W) in thin 1 x tw / th x 1 tiles.
#pragma once

struct UpTilePlan {
    int tiles_x, tiles_y, tw, th;      // main tiles
    int row_tiles, row_tw;             // position row gy = lim_y (0 tiles: the main tiles cover it)
    int col_tiles, col_th;             // position column gx = lim_x
    int lim_y, lim_x;                  // main tiles own positions gy < lim_y, gx < lim_x
    int total;
};

void conv16_up_tiles(int gh, int gw, int nw, int* tiles_x, int* tiles_y, int* tw, int* th);          // conv2d_bf16x3.hip

static inline UpTilePlan up_tile_plan(int H, int W, bool allow_edges) {
    UpTilePlan t;
    if (allow_edges && W >= 64 && W % 32 == 0 && H >= 64 && H % 8 == 0) {
        t.tw = 32; t.th = 8; t.tiles_x = W / 32; t.tiles_y = H / 8; t.lim_y = H; t.lim_x = W;
        t.row_tiles = (W + 1 + 146) / 147; t.row_tw = (W + 1 + t.row_tiles - 1) / t.row_tiles;       // 2 x (tw + 1) <= 297
        t.col_tiles = (H + 146) / 147; t.col_th = (H + t.col_tiles - 1) / t.col_tiles;
    } else {
        conv16_up_tiles(H + 1, W + 1, 8, &t.tiles_x, &t.tiles_y, &t.tw, &t.th);
        t.lim_y = H + 1; t.lim_x = W + 1; t.row_tiles = t.col_tiles = 0; t.row_tw = t.col_th = 1;
    }
    t.total = t.tiles_x * t.tiles_y + t.row_tiles + t.col_tiles;
    return t;
}

#ifdef __HIPCC__
// tile index -> origin, shape and the exclusive position bounds this tile may write (wave-uniform values)
__device__ __forceinline__ void up_tile_decode(const UpTilePlan& t, int tile_i, int& y0, int& x0, int& th, int& tw, int& end_y, int& end_x) {
    const int main_tiles = t.tiles_x * t.tiles_y;
    if (tile_i < main_tiles) {
        y0 = (tile_i / t.tiles_x) * t.th; x0 = (tile_i % t.tiles_x) * t.tw; th = t.th; tw = t.tw; end_y = t.lim_y; end_x = t.lim_x;
    } else if (tile_i < main_tiles + t.row_tiles) {
        y0 = t.lim_y; x0 = (tile_i - main_tiles) * t.row_tw; th = 1; tw = t.row_tw; end_y = t.lim_y + 1; end_x = t.lim_x + 1;
    } else {
        y0 = (tile_i - main_tiles - t.row_tiles) * t.col_th; x0 = t.lim_x; th = t.col_th; tw = 1; end_y = t.lim_y; end_x = t.lim_x + 1;
    }
}
#endif
