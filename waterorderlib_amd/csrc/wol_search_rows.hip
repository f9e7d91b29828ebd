// wol_search_rows.hip -- the proposal kernel, row-merged form (included by wol_api.hip).
//
// Same contract as k_search_fast (wol_search_fast.hip): per water the sorted slots of four proposed
// neighbours and the 4th fp32 key, certified later by k_finish_fast with the reference's fp64
// arithmetic.  What changes is how a tile is laid out in LDS, which is what the sweep's cost hangs on:
//
//   tile   = sx cells along x by ry cell rows along y in ONE z layer (long in x: whole rows if they fit);
//   staged = for every y row of the tile and its two y-halo rows, the three z rows (z-1, z, z+1) MERGED
//            into one array ordered by a fine x bin (WOL_XSUB bins per cell) -- a "triple".  The rows arrive
//            bin-sorted from k_sort_in_row: every water's fp32 record carries its cell, sub-bin and rank inside
//            the bin, cnt4[] the four sub-bin populations of every cell.  One pass, no atomics: a wave per triple
//            turns the three rows' cnt4 words into the first entry of every merged bin (wave scan) and the number
//            of waters of the rows below per bin; a wave per staged row then reads the row's three contiguous
//            pieces (left halo cell, the tile's cells, right halo cell; periodic images come out shifted because
//            the coordinates are built from staged cell numbers) and places every water at
//            first entry of its bin + waters of the rows below + its own rank.  Pair records as in k_search_fast.
//   sweep  = a water in row y looks at THREE contiguous windows, the bins [b-XSUB, b+XSUB] of triples
//            y-1, y, y+1 (b = its own fine bin): an x reach of at least one cell width plus the distance
//            to its bin wall, against nine spans of three whole cells before -- 25 % fewer candidates,
//            and three trip counts (wave maxima) per water instead of nine.
//
// The water's own record is left out by its position in the own triple (wol_rows_span), never by its distance: a
// second water at (nearly) the same place is a neighbour like any other.
#pragma once

// Two instantiations: 512 threads / 1280 staged pairs (49 KiB of LDS, three blocks per CU) and 768 threads / 1920
// pairs (71 KiB, two blocks per CU: the same 24 waves).  A tile lives in one cell layer, so small boxes quantise badly;
// the host picks per batch the instantiation whose tiles fill their lanes better: a 10 000-water box has 714 waters
// per layer -- two tiles of 357 for 512 lanes, one of 714 for 768 (fewer halo rows, half the per-tile work).  (A
// 384-thread form for still smaller boxes was measured slower: idle waves of a half-empty tile cost little.)
#define WOL_ROWS_BLOCK_A 512
#ifndef WOL_ROWS_CAPP_A
#define WOL_ROWS_CAPP_A 1280
#endif
#define WOL_ROWS_BLOCK_B 768
#define WOL_ROWS_CAPP_B 1920
#define WOL_ROWS_PADP 8        // head-room pairs behind the last triple (incl. the dummy pair)
#ifndef WOL_XSUB
#define WOL_XSUB 4             // fine x bins per cell
#endif
#define WOL_ROWS_MAXRY 30
#define WOL_ROWS_MAXTRIP (WOL_ROWS_MAXRY + 2)
#define WOL_ROWS_MAXROWS (3 * WOL_ROWS_MAXTRIP)
#define WOL_ROWS_MAXSX (252 / WOL_XSUB - 2)   // (sx + 2) * XSUB bins <= 252 (four bins per lane in the scan)
#ifndef WOL_ROWS_BINW
#define WOL_ROWS_BINW 704      // 32-bit words of bin counters: sum over the triples of ceil((bins + 1) / 2)
#endif
#define WOL_SLOT_BITS 20       // record id word = entry << 20 | sorted slot (frames of up to 2^20 waters)
#ifndef WOL_ROWS_ROUNDS
#define WOL_ROWS_ROUNDS 1      // sweep rounds per wave a tile is sized for (experiments: 2 = tiles twice as tall)
#endif
#ifndef WOL_ROWS_FILL
#define WOL_ROWS_FILL 97
#endif
#define WOL_ROWS_TARGET(block) (((block) * WOL_ROWS_FILL * WOL_ROWS_ROUNDS) / 100)
#define WOL_ROWS_CSCALE 0.0009765625f   // 2^-10
#ifndef WOL_ROWS_WPS
#define WOL_ROWS_WPS 6         // 80 VGPRs at most: three blocks (24 waves) per CU
#endif

// one candidate pair against the lane's water: two keys = squared distance in key units << 12 | entry number.
// Coordinates are staged in units of sqrt(2^-20 key range), so d^2 comes out in [0, 1] and saturates through the CLAMP
// modifier of its last fma; one more packed fma (x (2^20 - 1) + 2^23) turns it into the float whose mantissa IS the
// integer key (0 .. 2^20 - 1 = saturated).
// Records are split in two arrays of 16-byte halves, A = (x0, x1, y0, y1) and B = (z0, z1, id0, id1) per pair:
// a ds_read_b128 is served in groups of 16 lanes that conflict on (address / 16) mod 16, and consecutive pairs of
// one array fall on consecutive classes (interleaved 32-byte records would use every other class only).
template <int B_OFF>   // byte offset of array B behind array A
__device__ __forceinline__ void wol_rows_pair_keys(const char* __restrict__ rec_bytes, int off, wol_f2 xi, wol_f2 yi, wol_f2 zi,
                                                   unsigned& key0, unsigned& key1) {
    const float4 a = *reinterpret_cast<const float4*>(rec_bytes + off);
    const float4 b = *reinterpret_cast<const float4*>(rec_bytes + off + B_OFF);
    const wol_f2 dx = (wol_f2){a.x, a.y} - xi, dy = (wol_f2){a.z, a.w} - yi, dz = (wol_f2){b.x, b.y} - zi;
    wol_f2 u = dx * dx, uc, tt;
    u = __builtin_elementwise_fma(dy, dy, u);
    asm("v_pk_fma_f32 %0, %1, %1, %2 clamp" : "=v"(uc) : "v"(dz), "v"(u));            // min(d^2, 1)
    tt = __builtin_elementwise_fma(uc, (wol_f2){1048575.0f, 1048575.0f}, (wol_f2){8388608.0f, 8388608.0f});
    key0 = __builtin_amdgcn_alignbit(__float_as_uint(tt.x), __float_as_uint(b.z), WOL_SLOT_BITS);
    key1 = __builtin_amdgcn_alignbit(__float_as_uint(tt.y), __float_as_uint(b.w), WOL_SLOT_BITS);
}

// One window [pb, pb + np) of candidate pairs against the lane's water.  The pair holding the water's own record (byte
// offset self_off; -1 in the other triples' windows) is left out -- by POSITION, not by its distance: a second water
// within a hundredth of an Angstrom must stay a candidate (its key rounds to 0 like the water's own would) -- and is
// dealt with by the caller.
// Issue cost on gfx950 (tools/valu_probe.hip, profiles/r02_valu_issue_rates.txt): min / max / med3, shifts, compares,
// selects and the packed fp32 forms hold a SIMD for 4 cycles per wave64 instruction, plain fp32 add / mul / fma and
// integer add / and for 2.  So lanes past their window are masked off (v_cmp feeding EXEC) instead of being redirected to
// a dummy pair (v_cmp + v_cndmask).  (Measured and dropped: every lane sweeping the wave's trip count T inside its triple
// -- a superset of its window, no bounds test -- is slower, 1.41 against 1.38 ms: the LDS then serves 64 lanes in every
// iteration.)
template <int B_OFF>
__device__ __forceinline__ void wol_rows_span(const char* __restrict__ rec_bytes, int pb, int np, int self_off,
                                              wol_f2 xi, wol_f2 yi, wol_f2 zi, unsigned& k0,
                                              unsigned& k1, unsigned& k2, unsigned& k3, unsigned& k4) {
    // wave-uniform trip count = the longest window of the wave: a DPP reduction (shifts inside the rows of 16, two
    // row broadcasts, lane 63 holds the maximum) instead of six LDS-permute round trips; np >= 0 and all 64 lanes active
    int m = np;
    m = max(m, __builtin_amdgcn_update_dpp(0, m, 0x111, 0xf, 0xf, false));  // row_shr:1
    m = max(m, __builtin_amdgcn_update_dpp(0, m, 0x112, 0xf, 0xf, false));  // row_shr:2
    m = max(m, __builtin_amdgcn_update_dpp(0, m, 0x114, 0xf, 0xf, false));  // row_shr:4
    m = max(m, __builtin_amdgcn_update_dpp(0, m, 0x118, 0xf, 0xf, false));  // row_shr:8
    m = max(m, __builtin_amdgcn_update_dpp(0, m, 0x142, 0xa, 0xf, false));  // row_bcast:15
    m = max(m, __builtin_amdgcn_update_dpp(0, m, 0x143, 0xc, 0xf, false));  // row_bcast:31
    const int T = __builtin_amdgcn_readlane(m, 63);
    int off = pb * 16;
    const int end = (pb + np) * 16;
    for (int t = 0; t < T; ++t, off += 16) {
        if (off < end && off != self_off) {
            unsigned key0, key1;
            wol_rows_pair_keys<B_OFF>(rec_bytes, off, xi, yi, zi, key0, key1);
            wol_top5_insert(key0, k0, k1, k2, k3, k4);
            wol_top5_insert(key1, k0, k1, k2, k3, k4);
        }
    }
}

template <int WOL_ROWS_BLOCK, int WOL_ROWS_CAPP>
__global__ void __launch_bounds__(WOL_ROWS_BLOCK, WOL_ROWS_WPS)   // 6 waves per SIMD either way: 3 x 8 or 2 x 12 waves per CU
k_search_rows(FastArgs a) {
    __shared__ float4 rec[2 * (WOL_ROWS_CAPP + WOL_ROWS_PADP)];  // pair records: array A (x0,x1,y0,y1), then array B (z0,z1,id0,id1)
    __shared__ unsigned bins[WOL_ROWS_BINW];                     // per triple: first entry of every fine bin (16-bit), the total last
    __shared__ unsigned obins[WOL_ROWS_BINW];                    // per triple and cell: bytes = waters of the rows below in each sub-bin
    __shared__ int rsrc[WOL_ROWS_MAXROWS * 3];                   // staged row: first slot of its three pieces
    __shared__ unsigned short rcnt[WOL_ROWS_MAXROWS * 3];        // ... and their lengths
    __shared__ int rpre[WOL_ROWS_MAXROWS + 1];                   // staged waters before the row
    __shared__ int tbase[WOL_ROWS_MAXTRIP + 1];                  // first entry of the triple (even)
    __shared__ int qrow[WOL_ROWS_MAXRY + 2];                     // own rows -> first query number
    __shared__ int s_flag, s_next[2];   // (s_next: the tile number drawn by thread 0, two slots used in turn)
    const int tid = threadIdx.x;
    // (the wave number through readfirstlane: the compiler then keeps everything indexed by it -- the rows a wave stages, their
    //  table entries, the loops over them -- in scalar registers and scalar branches instead of 64 identical lanes under EXEC masks)
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int NW = WOL_ROWS_BLOCK / 64;
    const bool aff = a.xcd_affine != 0;
    const int fstep = aff ? 8 : 1;
    const int tstep = aff ? (int)(gridDim.x >> 3) : (int)gridDim.x;
    int f = aff ? (int)(blockIdx.x & 7) : 0;
    int tile = aff ? (int)(blockIdx.x >> 3) : (int)blockIdx.x;
    float* __restrict__ recf = reinterpret_cast<float*>(rec);
    float* __restrict__ recg = recf + 4 * (WOL_ROWS_CAPP + WOL_ROWS_PADP);   // array B
    // the wave's waters are dealt so that each of ds_read_b128's four lane groups holds 16 CONSECUTIVE waters
    // (their windows overlap: equal addresses broadcast, neighbouring ones fall on different banks)
    const int l5 = lane & 31;
    const bool g0 = (l5 < 4) || (l5 >= 12 && l5 < 16) || (l5 >= 20 && l5 < 28);
    const int lrank = (lane & 32) + (g0 ? (l5 < 4 ? l5 : l5 < 16 ? l5 - 8 : l5 - 12)
                                        : 16 + (l5 < 12 ? l5 - 4 : l5 < 20 ? l5 - 8 : l5 - 16));
#ifdef WOL_PHASE_STAMPS
    unsigned long long stamp_prev_ = __builtin_amdgcn_s_memtime();
#endif

    auto tiles_of = [&](int ff) -> int {
        if (ff >= a.F) return 0;
        const FrameGrid& gg = a.grids[ff];
        return (gg.maxshell < 0 || !gg.fast_ok) ? 0 : gg.row_ntiles[a.level[ff] & 0xFF];
    };
    // Per-frame constants stay in scalar registers from tile to tile (a block walks ~20 tiles of a frame before it
    // moves on): re-reading the frame's grid and re-deriving them per tile, on the vector unit of every wave, was a
    // sixth of the kernel's vector instructions.  Tile numbers are decoded with reciprocals (exact: operands < 2^22).
#define WOL_SGPR_I(x) __builtin_amdgcn_readfirstlane((int)(x))
#define WOL_SGPR_F(x) __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int((float)(x))))
    int f_cur = -1;
    int nx = 1, ny = 1, nz = 1, gsx = 1, gry = 1, gntx = 1, ntxy = 1, cell_base = 0, eps_units = 0, has_empty = 0;
    float inv_ntxy = 1.f, inv_gntx = 1.f, fxs = 0.f, fys = 0.f, fzs = 0.f, eps_a = 0.f, eps_b = 0.f;
    int nt_f = tiles_of(f);
    // Tile order.  Static (a.work == null): block b walks tiles b, b + grid, ... of its XCD's frames.  Dynamic: the blocks
    // of an XCD draw (frame, tile) numbers from that XCD's counter -- a block that starts late (the kernel shares the
    // CUs with the sort of the next batch) or runs on a busier CU then simply takes fewer tiles.  The next number is
    // drawn while the current tile is being worked on.
    const bool dyn = a.work != nullptr;
    int* __restrict__ ctr = dyn ? a.work + (aff ? 16 * (int)(blockIdx.x & 7) : 0) : nullptr;
    int w_next = 0, w_base = 0;   // next number drawn; first number of frame f (the numbers run through the XCD's frames)
    // s_flag ("this tile overflows its tables: exact path") is raised by several waves during a tile's table phase and read
    // by everybody behind that phase's last barrier; it is cleared where no wave can be raising it: here, and by thread 0
    // during the sweep (every thread has read it before the barrier in front of the sweep, and the next tile's first
    // barrier orders the clear before that tile's table phase)
    if (tid == 0) s_flag = 0;
    int slot = 0;   // which s_next slot the next hand-over uses: a slot is rewritten two hand-overs later, i.e. behind a barrier
                    // every thread passes after reading it
    if (dyn) {
        if (tid == 0) s_next[0] = atomicAdd(ctr, 1);
        __syncthreads();
        w_next = s_next[0];
        slot = 1;
    }
    for (;; tile += tstep) {
        if (dyn) {
            const int w = __builtin_amdgcn_readfirstlane(w_next);
            while (f < a.F && w >= w_base + nt_f) { w_base += nt_f; f += fstep; nt_f = tiles_of(f); }
            if (f >= a.F) break;
            tile = w - w_base;
        } else {
            while (f < a.F && tile >= nt_f) { tile -= nt_f; f += fstep; nt_f = tiles_of(f); }
            if (f >= a.F) break;
        }
        if (f != f_cur) {
            const FrameGrid& g = a.grids[f];
            const int levw = a.level[f];
            const int lev = levw & 0xFF;
            has_empty = WOL_SGPR_I(a.tile_any ? (levw >> 8) & 1 : 0);
            nx = WOL_SGPR_I(g.nc[0]); ny = WOL_SGPR_I(g.nc[1]); nz = WOL_SGPR_I(g.nc[2]);
            gsx = WOL_SGPR_I(g.row_sx[lev]); gry = WOL_SGPR_I(g.row_ry[lev]);
            gntx = WOL_SGPR_I(g.row_ntx[lev]);
            ntxy = WOL_SGPR_I(gntx * (int)g.row_nty[lev]);
            cell_base = WOL_SGPR_I(g.cell_base);    // (a batch's cell tables stay below 2^31 entries)
            eps_units = WOL_SGPR_I(g.row_eps_units);
            inv_ntxy = WOL_SGPR_F(1.0f / (float)ntxy); inv_gntx = WOL_SGPR_F(1.0f / (float)gntx);
            // the records hold coordinates in units of 2^-10 sqrt(key unit) (an exact power-of-two rescaling of the
            // sorted fp32 copy): squared distances then come out as fractions of the key range and saturate through CLAMP
            fxs = WOL_SGPR_F((float)(g.cw[0] * (double)g.scale) * WOL_ROWS_CSCALE);
            fys = WOL_SGPR_F((float)(g.cw[1] * (double)g.scale) * WOL_ROWS_CSCALE);
            fzs = WOL_SGPR_F((float)(g.cw[2] * (double)g.scale) * WOL_ROWS_CSCALE);
            eps_a = WOL_SGPR_F(g.row_eps_a); eps_b = WOL_SGPR_F(g.row_eps_b);
            f_cur = f;
        }
        const int cz = (int)(((float)tile + 0.5f) * inv_ntxy);
        const int trem = tile - cz * ntxy;
        const int ty = (int)(((float)trem + 0.5f) * inv_gntx);
        const int tx = trem - ty * gntx;
        // a tile row without any water (k_pick_level's flags; frames that have such rows say so in their level word, so
        // a full box pays nothing here): nothing to propose, nothing to write
        // (tile = (cz * nty + ty) * ntx + tx: the flag's index is the tile number without its tx.  The flag is fetched before
        // the next tile number is drawn: a wait for it must not have to wait for that atomic's round trip as well)
        int tile_flag = 1;
        if (has_empty) tile_flag = a.tile_any[(int64_t)f * a.tile_any_stride + (int)(((float)(tile - tx) + 0.5f) * inv_gntx)];
        if (dyn && tid == 0) w_next = atomicAdd(ctr, 1);   // (in flight until the end of this tile)
        if (__builtin_amdgcn_readfirstlane(tile_flag) == 0) {   // nothing in this tile row: on to the next number
            if (dyn) {
                if (tid == 0) s_next[slot] = w_next;
                __syncthreads();
                w_next = s_next[slot];
                slot ^= 1;
            }
            continue;
        }
        const int x0 = tx * gsx, sxe = min(gsx, nx - x0);
        const int y0 = ty * gry, re = min(gry, ny - y0);
        const int ntrip = re + 2, nrows = 3 * ntrip;
        const int nb = (sxe + 2) * WOL_XSUB, nbw = (nb + 2) >> 1;   // bins, words per triple (nb + 1 halves)
        const unsigned short* __restrict__ bins16 = reinterpret_cast<const unsigned short*>(bins);
        const float4* __restrict__ sp32 = a.sorted32 + (int64_t)f * a.n;
        const int* __restrict__ cs = a.cell_start + cell_base;
        const int64_t frame_off = (int64_t)f * a.n;
        const float xc = 0.5f * (float)(sxe + 2), yc = 0.5f * (float)(re + 2);

        // the previous tile must be done with the LDS tables: with dynamic tile numbers the barrier of the hand-over at its end
        // (or of the kernel's first draw) is that point already
        if (!dyn) __syncthreads();
        WOL_STAMP(0);
        // ---- phase 1: tables.  Threads 0 .. nrows-1: the three pieces (left halo cell, the tile's cells, right halo cell)
        //      of one staged row each, from cell_start.  The other waves, one per triple: the sub-bin populations of the
        //      triple's cells (cnt4: four bytes per cell, written by the in-row sort) of its three z rows -> first entry of
        //      every fine bin of the merged triple (bins[], 16-bit) and, per bin, how many waters of the rows below
        //      precede a row's own (obins[]: bytes).  The rows are bin-sorted and every record carries its rank inside its
        //      bin, so these tables place a staged water without any counting pass. ----
        if (tid < nrows) {
            const int t = (int)(((float)tid + 0.5f) * (1.0f / 3.0f)), dzk = tid - 3 * t;
            const int gz = wol_wrap_cell(cz + dzk - 1, nz), gy = wol_wrap_cell(y0 + t - 1, ny);
            const int rb = (gz * ny + gy) * nx;
            const int cl = wol_wrap_cell(x0 - 1, nx), cr = wol_wrap_cell(x0 + sxe, nx);
            const int sl = cs[rb + cl], el = cs[rb + cl + 1];
            const int sb = cs[rb + x0], eb = cs[rb + x0 + sxe];
            const int sr = cs[rb + cr], er = cs[rb + cr + 1];
            rsrc[3 * tid] = sl; rsrc[3 * tid + 1] = sb; rsrc[3 * tid + 2] = sr;
            rcnt[3 * tid] = (unsigned short)min(el - sl, 65535);
            rcnt[3 * tid + 1] = (unsigned short)min(eb - sb, 65535);
            rcnt[3 * tid + 2] = (unsigned short)min(er - sr, 65535);
        }
        {
            // (the row table above keeps (nrows + 63) / 64 waves busy -- one, for tiles of up to 19 rows: the other seven then
            //  take one triple each on the benchmark box's 32 x 5-cell tiles, where six waves had one of them do two in turn)
            const int W0 = (nrows + 63) >> 6;
            const int ncs = sxe + 2;                             // staged cells of a row (<= 63: one lane each)
            for (int t = wave - W0; t >= 0 && t < ntrip; t += NW - W0) {
                unsigned w0 = 0u, w1 = 0u, w2 = 0u;
                if (lane < ncs) {
                    const int gx = lane == 0 ? wol_wrap_cell(x0 - 1, nx) : lane == ncs - 1 ? wol_wrap_cell(x0 + sxe, nx) : x0 + lane - 1;
                    const int gy = wol_wrap_cell(y0 + t - 1, ny);
                    const unsigned* __restrict__ c4 = a.cnt4 + cell_base + gx;
                    w0 = c4[(wol_wrap_cell(cz - 1, nz) * ny + gy) * nx];
                    w1 = c4[(cz * ny + gy) * nx];
                    w2 = c4[(wol_wrap_cell(cz + 1, nz) * ny + gy) * nx];
                }
                // byte-wise sums stay inside their bytes while every population is below 64; denser bins (and
                // saturated counts) send the tile to the exact path
                if ((w0 | w1 | w2) & 0xC0C0C0C0u) s_flag = 1;
                const unsigned w01 = w0 + w1, wt = w01 + w2;
                const int c0 = (int)(wt & 0xFFu), c1 = (int)((wt >> 8) & 0xFFu), c2 = (int)((wt >> 16) & 0xFFu), c3 = (int)(wt >> 24);
                const int s = (c0 + c1) + (c2 + c3);
                const int inc = wol_wave_scan_incl(s);
                const int run = inc - s;
                unsigned* __restrict__ bt = bins + t * nbw;
                unsigned* __restrict__ ob = obins + t * nbw;
                if (lane < ncs) {
                    bt[2 * lane] = (unsigned)run | ((unsigned)(run + c0) << 16);
                    bt[2 * lane + 1] = (unsigned)(run + c0 + c1) | ((unsigned)(run + c0 + c1 + c2) << 16);
                    ob[2 * lane] = w0;        // waters of the z-1 row in the bin: they precede the z row's
                    ob[2 * lane + 1] = w01;   // ... and the z-1 and z rows' precede the z+1 row's
                }
                if (lane == ncs - 1) bt[2 * ncs] = (unsigned)inc;   // bin nb holds the total
            }
        }
        __syncthreads();
        if (wave == 0) {           // staged waters before each row (nrows <= 96: two rounds of 64)
            int run = 0;
            for (int r0 = 0; r0 < nrows; r0 += 64) {
                const int r = r0 + lane;
                const int len = (r < nrows) ? (int)rcnt[3 * r] + (int)rcnt[3 * r + 1] + (int)rcnt[3 * r + 2] : 0;
                const int inc = wol_wave_scan_incl(len);
                if (r < nrows) rpre[r] = run + inc - len;
                run += __builtin_amdgcn_readlane(inc, 63);
            }
            if (lane == 0) {
                rpre[nrows] = run;
                if (run + ntrip > 2 * WOL_ROWS_CAPP) s_flag = 1;
            }
        } else if (wave == 1) {    // own rows (centre z row of triples 1..re, the tile's own cells)
            const int c = (lane < re) ? (int)rcnt[3 * (3 * (lane + 1) + 1) + 1] : 0;
            const int inc = wol_wave_scan_incl(c);
            if (lane < re) qrow[lane + 1] = inc;
            if (lane == 0) qrow[0] = 0;
        } else if (wave == 2) {    // first entry of every triple (even): its size is known from the row table already
            int tot = 0;
            if (lane < ntrip)
#pragma unroll
                for (int u = 0; u < 9; ++u) tot += (int)rcnt[9 * lane + u];
            // the sub-bin populations must add up to the cell populations (they saturate at 255 per bin)
            if (lane < ntrip && (int)(bins[lane * nbw + 2 * (sxe + 2)] & 0xFFFFu) != tot) s_flag = 1;
            const int len = (tot + 1) & ~1;
            const int inc = wol_wave_scan_incl(len);
            if (lane < ntrip) {
                tbase[lane] = inc - len;
                if ((tot & 1) && inc <= 2 * WOL_ROWS_CAPP) {   // complete the triple's last pair with a far-away dummy
                    const int e = inc - len + tot;
                    const int o = (e >> 1) * 4 + 1;
                    recf[o] = 1.0e7f; recf[o + 2] = 1.0e7f; recg[o] = 1.0e7f;
                    recg[o + 2] = __uint_as_float((unsigned)e << WOL_SLOT_BITS);
                }
            }
            if (lane == ntrip - 1) tbase[ntrip] = inc;
        }
        __syncthreads();
        const bool overflow = s_flag != 0;
        const int total_e = tbase[ntrip];
        WOL_STAMP(1);
        // ---- phase 2: every water to its place, ONE pass: entry = first entry of the triple's bin + the waters of the rows
        //      below in that bin + its rank inside the bin.  A wave takes whole rows; the fetch of the next 64 waters is in
        //      flight while the current 64 are placed. ----
#ifndef WOL_EXP_NO_PLACE   // (ablation build with -DWOL_EXP_NO_PLACE: instruction counts only, results are wrong)
        if (!overflow) {
            const unsigned char* __restrict__ ob8 = reinterpret_cast<const unsigned char*>(obins);
            int r = wave, i0 = 0;
            // the current row's pieces and everything else that is the same for all of its waters (wave-uniform: kept in
            // scalar registers, recomputed when the wave moves to its next row)
            int c0 = 0, c1 = 0, cnt = 0, d0 = 0, d1 = 0, d2 = 0, tb = 0, binbase = 0, obbase = -1;
            float offy = 0.f, offz = 0.f;
#define WOL_ROW_PARAMS(rr)                                                                                       \
            do {                                                                                                 \
                const int t_ = (rr) / 3, dzk_ = (rr) - 3 * t_;                                                   \
                c0 = WOL_SGPR_I(rcnt[3 * (rr)]); c1 = WOL_SGPR_I(rcnt[3 * (rr) + 1]);                            \
                cnt = c0 + c1 + WOL_SGPR_I(rcnt[3 * (rr) + 2]);                                                  \
                d0 = WOL_SGPR_I(rsrc[3 * (rr)]); d1 = WOL_SGPR_I(rsrc[3 * (rr) + 1]) - c0;                       \
                d2 = WOL_SGPR_I(rsrc[3 * (rr) + 2]) - c0 - c1;                                                   \
                tb = WOL_SGPR_I(tbase[t_]);                                                                      \
                binbase = 2 * t_ * nbw;                                 /* first 16-bit bin entry of the triple */ \
                obbase = dzk_ == 0 ? -1 : (t_ * nbw + (dzk_ - 1)) * 4;  /* byte table of the rows below */       \
                offy = WOL_SGPR_F(((float)t_ - yc) * fys); offz = WOL_SGPR_F((float)(dzk_ - 1) * fzs);           \
            } while (0)
            while (r < nrows) { WOL_ROW_PARAMS(r); if (cnt > 0) break; r += NW; }   // first non-empty row of this wave
            float4 vn;
            vn.x = vn.y = vn.z = vn.w = 0.f;
            // (index records: the id word names the water's ORIGINAL index, fetched with its record)
            const int* __restrict__ sidx = a.sidx ? a.sidx + frame_off : nullptr;
            int idn = 0;
            int srcn = lane + (lane < c0 ? d0 : lane >= c0 + c1 ? d2 : d1);
            if (r < nrows && lane < cnt) { vn = sp32[srcn]; idn = sidx ? sidx[srcn] : srcn; }
            while (r < nrows) {
                // the current chunk of 64 waters
                const float4 v = vn;
                const int src = idn, i = i0 + lane;
                const int cc0 = c0, cc1 = c1, ccnt = cnt, ctb = tb, cbin = binbase, cob = obbase;
                const float coffy = offy, coffz = offz;
                // advance to the next chunk and start its fetch
                i0 += 64;
                if (i0 >= cnt) {
                    i0 = 0;
                    do { r += NW; if (r < nrows) WOL_ROW_PARAMS(r); } while (r < nrows && cnt == 0);
                }
                if (r < nrows) {
                    const int in_ = i0 + lane;
                    srcn = in_ + (in_ < c0 ? d0 : in_ >= c0 + c1 ? d2 : d1);
                    if (in_ < cnt) { vn = sp32[srcn]; idn = sidx ? sidx[srcn] : srcn; }
                }
                // place the current chunk
                if (i < ccnt) {
                    const bool p0 = i < cc0, p2 = i >= cc0 + cc1;
                    const unsigned tag = __float_as_uint(v.w);
                    const int cx = (int)(tag & 0xFFu), sub = (int)((tag >> 8) & 3u), rank = (int)(tag >> 10);
                    const int sc = p0 ? 0 : p2 ? sxe + 1 : cx - x0 + 1;
                    const int bin = sc * WOL_XSUB + sub;
                    const int below = cob < 0 ? 0 : (int)ob8[cob + 8 * sc + sub];
                    const int e = ctb + (int)bins16[cbin + bin] + below + rank;
                    const int o = (e >> 1) * 4 + (e & 1);
                    recf[o] = __builtin_fmaf(v.x, WOL_ROWS_CSCALE, ((float)sc - xc) * fxs);
                    recf[o + 2] = __builtin_fmaf(v.y, WOL_ROWS_CSCALE, coffy);
                    recg[o] = __builtin_fmaf(v.z, WOL_ROWS_CSCALE, coffz);
                    recg[o + 2] = __uint_as_float(((unsigned)e << WOL_SLOT_BITS) | (unsigned)src);
                }
            }
#undef WOL_ROW_PARAMS
            if (tid < WOL_ROWS_PADP) {   // head-room pairs (incl. the dummy pair): nothing near
                const int p = (total_e >> 1) + tid;
                float4 far;
                far.x = far.y = far.z = far.w = 1.0e7f;
                rec[p] = far;
                far.z = far.w = __int_as_float(0);
                rec[p + WOL_ROWS_CAPP + WOL_ROWS_PADP] = far;
            }
        }
#endif
        __syncthreads();
        WOL_STAMP(4);
        if (tid == 0) s_flag = 0;   // (for the next tile: see the note at the kernel's first clear)
        // ---- sweep: one lane per own water, three windows ----
        const int nq = qrow[re];
        for (int qb = wave * 64; qb < nq; qb += WOL_ROWS_BLOCK) {
            // every lane of the wave stays in the sweep (the trip-count butterfly needs all 64): lanes past the
            // tile's last water shadow that water with empty windows and store nothing
            const bool live = qb + lrank < nq;
            const int qi = live ? qb + lrank : nq - 1;
            int j = 0;  // own row: largest j with qrow[j] <= qi (re <= 30)
#pragma unroll
            for (int step = 16; step >= 1; step >>= 1)
                if (j + step < re && qrow[j + step] <= qi) j += step;
            const int self_slot = rsrc[3 * (3 * (j + 1) + 1) + 1] + (qi - qrow[j]);
            int4 prop;
            prop.x = prop.y = prop.z = prop.w = 0;
            int kd3 = -1;
            if (!overflow) {
                const float4 me = sp32[self_slot];
                const unsigned mtag = __float_as_uint(me.w);
                const int sc = (int)(mtag & 0xFFu) - x0 + 1;
                const int b = sc * WOL_XSUB + (int)((mtag >> 8) & 3u);
                // the water's own entry in its triple, as the placement pass computed it (centre z row: the z-1 row's
                // waters of the bin come first)
                const int e_self = tbase[j + 1] + (int)bins16[2 * (j + 1) * nbw + b] +
                                   (int)reinterpret_cast<const unsigned char*>(obins)[(j + 1) * nbw * 4 + 8 * sc + (int)((mtag >> 8) & 3u)] +
                                   (int)(mtag >> 10);
                const int self_off = (e_self >> 1) * 16;
                const float mx = __builtin_fmaf(me.x, WOL_ROWS_CSCALE, ((float)sc - xc) * fxs);
                const float my = __builtin_fmaf(me.y, WOL_ROWS_CSCALE, ((float)(j + 1) - yc) * fys), mz = me.z * WOL_ROWS_CSCALE;
                const wol_f2 xi = {mx, mx}, yi = {my, my}, zi = {mz, mz};
                unsigned k0 = 0xFFFFFFFFu, k1 = 0xFFFFFFFFu, k2 = 0xFFFFFFFFu, k3 = 0xFFFFFFFFu, k4 = 0xFFFFFFFFu;
                // the three windows as (pairs << 16 | triple << 12 | first pair); each lane sweeps its longest first: the
                // loops' trip counts are wave maxima, and maxima of sorted lengths add up to less (38.3 instead of 40.5 pairs)
                unsigned win[3];
#pragma unroll
                for (int dt = 0; dt < 3; ++dt) {
                    const int t = j + dt;
                    const unsigned short* __restrict__ bt = bins16 + 2 * t * nbw;
                    const int lo = tbase[t] + (int)bt[b - WOL_XSUB], hi = tbase[t] + (int)bt[b + WOL_XSUB + 1];
                    const int pb = lo >> 1, pe = live ? (hi + 1) >> 1 : pb;
                    win[dt] = ((unsigned)(pe - pb) << 16) | ((unsigned)dt << 12) | (unsigned)pb;
                }
#ifndef WOL_EXP_NO_WINSORT
                {
                    const unsigned a01 = max(win[0], win[1]), b01 = min(win[0], win[1]);
                    const unsigned c2 = max(b01, win[2]);
                    win[2] = min(b01, win[2]); win[0] = max(a01, c2); win[1] = min(a01, c2);
                }
#endif
#pragma unroll
                for (int dt = 0; dt < 3; ++dt) {
#ifdef WOL_EXP_NO_SWEEP   // ablation build: timing only, results are wrong
                    k0 ^= win[dt];
#else
                    const int wt = (int)((win[dt] >> 12) & 3u);     // which triple this window lies in
                    wol_rows_span<16 * (WOL_ROWS_CAPP + WOL_ROWS_PADP)>(reinterpret_cast<const char*>(rec), (int)(win[dt] & 0xFFFu), (int)(win[dt] >> 16),
                                  wt == 1 ? self_off : -1,
                                  xi, yi, zi, k0, k1, k2, k3, k4);
#endif
                }
#ifndef WOL_EXP_NO_SWEEP
#ifndef WOL_EXP_NO_PARTNER
                {   // the other water of the own pair (the sweeps skipped the pair as a whole)
                    unsigned key0, key1;
                    wol_rows_pair_keys<16 * (WOL_ROWS_CAPP + WOL_ROWS_PADP)>(reinterpret_cast<const char*>(rec), self_off, xi, yi, zi, key0, key1);
                    wol_top5_insert((e_self & 1) ? key0 : key1, k0, k1, k2, k3, k4);
                }
#endif
#endif
                const unsigned d3 = k3 >> WOL_ID_BITS, d4 = k4 >> WOL_ID_BITS;
                const unsigned sat = (1u << (32 - WOL_ID_BITS)) - 1u - (unsigned)eps_units;
                // the proposal stands when the 5th key is separated from the 4th by more than both keys' fp32 error
                // (bound evaluated at the 5th distance, the larger of the two)
                const float eps4 = eps_a * __builtin_sqrtf((float)d4) + eps_b;
                if (k3 != 0xFFFFFFFFu && d3 < sat && (float)(d4 - d3) > 2.0f * eps4) {
                    const unsigned m = (1u << WOL_ID_BITS) - 1u, sm = (1u << WOL_SLOT_BITS) - 1u;
                    auto slot_of = [&](unsigned key) -> int {
                        const unsigned e = key & m;
                        return (int)(__float_as_uint(recg[(e >> 1) * 4 + 2 + (e & 1)]) & sm);
                    };
                    prop.x = slot_of(k0); prop.y = slot_of(k1); prop.z = slot_of(k2); prop.w = slot_of(k3);
                    {   // in ascending order of the slots (the certification does not care about the order): the k-th gathers
                        // of a wave's 64 neighbouring waters then fall into the same few row segments of the sorted array
                        // (certification 664 -> 654 us per step; no more than that: it is the number of lines, not their spread)
                        int t0 = min(prop.x, prop.y), t1 = max(prop.x, prop.y), t2 = min(prop.z, prop.w), t3 = max(prop.z, prop.w);
                        prop.x = min(t0, t2); prop.w = max(t1, t3);
                        const int u = max(t0, t2), v = min(t1, t3);
                        prop.y = min(u, v); prop.z = max(u, v);
                    }
                    kd3 = (int)d3;
                }
            }
            if (live) {
                a.prop[frame_off + self_slot] = prop;
                a.prop_key[frame_off + self_slot] = kd3;
            }
        }
        WOL_STAMP(5);
        if (dyn) {
            if (tid == 0) s_next[slot] = w_next;
            __syncthreads();
            w_next = s_next[slot];
            slot ^= 1;
        }
    }
}
