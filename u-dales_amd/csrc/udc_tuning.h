// Tuning constants of the kernels' launch shapes.  Through round 4 each was an environment knob (UDC_XPAD, UDC_SPEC_PAD, UDC_FFT_L / _C,
// UDC_NAT_L / _C, UDC_MOM_KC, UDC_SCALAR_KC, UDC_CLOSURE_PERCU); every scan ended on one value, kept here with the measurement that
// chose it (profiles/HISTORY.md has the scans).  Nothing in this header changes a result.
#pragma once

namespace tune {

// Row padding of every 3-D field (Geo.sy = nx + XPAD): one 128-B line per row where power-of-two rows of 4 KiB and more put all rows of
// a tile column on the same few L2 channels and sets (closure 3.2 -> 2.55 ms, momentum 5.2 -> 4.95 ms at 1024 x 512 x 512, +1 % at
// nx = 512, nothing at 256: profiles/r01).
constexpr int XPAD_DOUBLES = 16;
constexpr int XPAD_MIN_NX = 512, XPAD_NX_MULTIPLE = 256;
inline int row_padding(int nx) { return (nx >= XPAD_MIN_NX && nx % XPAD_NX_MULTIPLE == 0) ? XPAD_DOUBLES : 0; }

// k-chunks of the marching sweeps: each workgroup pays a 3-plane prologue and the chip runs 256 x per_cu workgroups at a time; the
// chunk minimises rounds x (kc + 3) (pick_kc).  Workgroups per CU: momentum 3 (51.7 KB of LDS), closure 4, or 5 from 1024 tiles on
// (32.6 KB; 512 x 512 x 256 0.634 -> 0.592 ms, 256^3 0.172 against 0.19: profiles/r02), kappa sweep: its own rule in udc_scalar_lds.hip.
constexpr int MOM_PER_CU = 3;
#ifndef UDC_CLOSURE5_MIN_TILES
#define UDC_CLOSURE5_MIN_TILES 1024
#endif
inline int closure_per_cu(int tiles) { return tiles >= UDC_CLOSURE5_MIN_TILES ? 5 : 4; }

// Line transforms (udc_fft.hip).  Stockham x kernels of the slab path: rows per workgroup as long as four workgroups fit a CU's LDS,
// at least 4 (L = 4 at nx = 1024: 18.4 against 20.3 ms per substep with 8, profiles/r03/fft_threads_scan.txt); Stockham y kernels:
// 8 columns.  One-GPU forward x kernel: 4 rows (256^3 0.085 against 0.088 / 0.097 ms with 2 / 8; profiles/r03/nat_l_scan.json),
// register y pass: 8 columns.
constexpr int FFT_X_LDS_BUDGET = 40000, FFT_X_MIN_ROWS = 4, FFT_Y_COLS = 8;
constexpr int NAT_X_ROWS = 4, NAT_Y_COLS = 8;

// Thomas solve with one system per thread (the slab ranks' lines shorter than 256, UDC_THOMAS_PAIR=0): sixteen levels per thread
// instead of eight above this many levels (one rank's slab of eight of 1024 x 512 x 512: 150 -> 134 us; the same grid unpaired on one GPU
// 1.27 -> 1.15 ms; no difference at 256 levels: profiles/r05/thomas_scan_ab.txt).
constexpr int THOMAS_SL16_ABOVE = 256;

}  // namespace tune
