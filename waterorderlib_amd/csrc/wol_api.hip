// wol_api.hip -- host side of libwol.so: the C ABI declared in include/wol.h.
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -shared wol_api.hip -o libwol.so
#include "wol_kernels.hip"
#include "wol_search_fast.hip"
#include "wol_search_rows.hip"
#include "wol_shell.hip"
#include "wol_shell_traj.hip"

#include <dlfcn.h>
#include <rccl/rccl.h>   // types and enums only: the library is resolved at run time (wol_comm_*), never linked

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <atomic>
#include <string>
#include <thread>
#include <vector>

#include "../../include/wol.h"

namespace {

thread_local std::string g_err;

int fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

#define HIP_TRY(expr)                                                                         \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess)                                                                 \
            return fail(e_ == hipErrorOutOfMemory ? WOL_ERR_NOMEM : WOL_ERR_HIP, "%s: %s",    \
                        #expr, hipGetErrorString(e_));                                        \
    } while (0)

#define WOL_TRY(expr)            \
    do {                         \
        int rc_ = (expr);        \
        if (rc_ != WOL_OK) return rc_; \
    } while (0)

struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    int ensure(size_t bytes) {
        if (bytes <= cap) return WOL_OK;
        if (p) { (void)hipFree(p); p = nullptr; cap = 0; }
        size_t want = bytes + bytes / 8 + 256;
        HIP_TRY(hipMalloc(&p, want));
        cap = want;
        return WOL_OK;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
    template <typename T> T* as() const { return reinterpret_cast<T*>(p); }
};

struct PinBuf {
    void* p = nullptr;
    size_t cap = 0;
    int ensure(size_t bytes) {
        if (bytes <= cap) return WOL_OK;
        if (p) { (void)hipHostFree(p); p = nullptr; cap = 0; }
        size_t want = bytes + bytes / 8 + 256;
        HIP_TRY(hipHostMalloc(&p, want, hipHostMallocDefault));
        cap = want;
        return WOL_OK;
    }
    void release() { if (p) (void)hipHostFree(p); p = nullptr; cap = 0; }
    template <typename T> T* as() const { return reinterpret_cast<T*>(p); }
};

}  // namespace

#define WOL_NSETS 3   // batches in flight in the trajectory path (sort / proposal / certification)
struct wol_ctx {
    int device = 0;
    int num_cus = 256;
    size_t sort_lds_set = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    // options
    double cell_width = 4.5;
    int frames_per_batch = 0;
    int path = 0;
    bool time_kernels = false;
    bool lds_sort = true;
    // workspace
    // Three batches are in flight in the trajectory path: the sort of batch k+2 (memory-bound) on `sort_stream`, the
    // proposal kernel of batch k+1 (bound by the vector issue rate) on `search_stream` and the certification of batch k
    // (gathers + fp64) on `stream`, the stream the caller sees.  Each batch owns one set of intermediate arrays; events
    // hand the sets from stage to stage.
    struct SortSet {
        DevBuf sorted, sorted32, cells, grids, level, cnt4, prop, prop_key, work, tile_any;
        hipEvent_t sorted_ev = nullptr;    // recorded on the sort stream when the set is filled
        hipEvent_t searched_ev = nullptr;  // recorded on the search stream when the proposals are written
        hipEvent_t rows_ev = nullptr, level_ev = nullptr;   // first sort level done / density level picked (beside the second level)
        hipEvent_t free_ev = nullptr;      // recorded on `stream` when its last reader is done
        bool used = false;
    } sets[WOL_NSETS];
    int next_set = 0;
    hipStream_t sort_stream = nullptr, search_stream = nullptr;
    hipEvent_t in_ev = nullptr;
    hipEvent_t scratch_ev = nullptr;   // recorded behind every sort: the sorts share scratch arrays whichever stream they run on
    bool search_rows = true;  // proposal kernel: row-merged tiles (wol_search_rows.hip); false = 3-D tiles (wol_search_fast.hip)
    bool overlap = false;  // measured on MI355X: the sort and the search contend for the same CU slots and
                           // memory latency, running them side by side gains nothing; kept as an opt-in
    bool inputs_resident = false;
    size_t sort_rows_lds_set = 0;
    size_t in_row_static_lds[6] = {0, 0, 0, 0, 0, 0}, in_row_dyn_limit[6] = {0, 0, 0, 0, 0, 0};   // k_sort_in_row<2> / <4> (+ 2: float4 records, + 4: index records): static LDS, dynamic LDS allowed so far
    int idx_sort = 0;        // float64 frames: index records through the sort, fp64 from the source frame (option "idx_sort"; measured slower: off)
    int rec32 = 1;           // float32 frames keep 16-byte records through the sort and the certification (option "rec32"; 0 = promote in the first level)
    int last_rows_block = 0;
    int last_rec_bytes = 0;
    int force_rows_block = 0;  // 0 = per batch by lane fill; 512 / 768 = that k_search_rows instantiation (tests, tuning)
    int sort_parts = 0;      // > 0: blocks per frame of the streaming sort's first level (0 = automatic)
    int dyn_tiles = 1;       // 1: the proposal kernel's blocks draw their tiles from per-XCD counters (default); 0: static tile order
    int seg_sort = 1;        // first sort level in its single-pass segment form: 1 = for frames of 8 chunks (24 576 waters) and
                             // more (smaller frames sort faster through row buckets: their second read comes out of L2), 2 = whenever it applies, 0 = never
    int stages = 2;          // pipe depth of the trajectory path: 2 = sort | proposal + certification, 3 = sort | proposal | certification
    int search_margin = 0;   // blocks per CU the proposal / finish kernels leave free (room for an overlapped sort)
    bool cos_traj_lds_set = false;
    bool skip_empty = true;         // the proposal kernel skips tile rows without any water (option "skip_empty_tiles")
    bool pair_tiled = true;         // RadialDistSame through the LDS-tiled pair kernel (option "pair_tiled"; 0 = the lane-per-water walk)
    bool check_props = false;       // diagnostic: poison the proposal arrays, count what the proposal kernel writes (k_check_props)
    DevBuf prop_counters;           // {keys written, proposals naming a slot outside the frame} since the option was switched on
    int64_t prop_expected = 0;      // waters the row kernel was given since then
    SortSet* last_set = nullptr;
    DevBuf seg_tab;   // segment form of the first sort level: start of every row inside every chunk (16 bit)
    DevBuf cellid, rank, qbuf, in_stage, out_stage, hist_stage, small, stat, layer_off, sort_tmp, slow_list, slow_count, stamps, scan_tot;
    PinBuf grids_host;
    DevBuf cvt_in, cvt_out[4], xchg, stat_part;
    int stat_slots = 0;   // slots per frame of stat_part in the last batch (0: unused)   // float <-> double conversion staging of the typed per-frame calls; exchange buffer of the sharded call
    // host-trajectory pipeline (wol_q_traj with WOL_MEM_HOST): two staging sets; the copy of batch k+1 runs on
    // copy_stream under batch k's kernels
    DevBuf host_stage[2];
    PinBuf pack_stage[2];           // pinned staging of the host-side O-row packer
    int host_pack = 2;              // what host frames cross the link as: 0 whole frames, 1 packed O rows, 2 packed + float32 when exact
    int host_threads = 0;           // packing threads (0 = min(16, hardware threads))
    DevBuf pop_off, pop_pts, pop_stat, pop_hist, pop_out, pop_res;   // populations of a trajectory call; results of the shell trajectory calls
    hipStream_t copy_stream = nullptr;
    hipEvent_t copied_ev[2] = {nullptr, nullptr}, consumed_ev[2] = {nullptr, nullptr};
    double last_h2d_bytes = 0.0;   // bytes the last host-trajectory call moved over the link (diagnostic counter)
    hipEvent_t grid_ev = nullptr;   // marks completion of the last pinned->device grid upload
    hipStream_t grid_stream = nullptr;
    bool shell_dirty = false;       // a per-frame shell call (cos_angs, lsi, ...) used sets[0] on `stream` since the last batch
    hipEvent_t shell_ev = nullptr;
    bool grid_copy_pending = false;
    // dominant-kernel timing
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_used, ev_free;
    double ms_total = 0.0;
    int64_t launches = 0;
};

namespace {

Box3 make_box(const double* box) {
    Box3 b;
    for (int k = 0; k < 3; ++k) {
        b.L[k] = box[k];
        b.ib[k] = (box[k] >= 0.0) ? 1.0 / box[k] : 0.0;  // waterlib.f90:41
    }
    return b;
}

// Tile tables of the proposal kernels: they depend on the cell counts and the number of waters only, not on the box
// lengths themselves, so a trajectory whose box breathes (NPT) reuses them from frame to frame; only the scalars
// (cell widths, key scale, error bounds) are recomputed per frame.
struct TileTables {
    int nlev = 0;
    float lev_wpc[WOL_NLEV];
    short lev_sx[WOL_NLEV], lev_ry[WOL_NLEV], lev_sz[WOL_NLEV];
    short lev_ntx[WOL_NLEV], lev_nty[WOL_NLEV], lev_ntz[WOL_NLEV];
    int lev_ntiles[WOL_NLEV];
    int ntiles = 0;
    short row_sx[WOL_NLEV], row_ry[WOL_NLEV], row_ntx[WOL_NLEV], row_nty[WOL_NLEV];
    int row_ntiles[WOL_NLEV];
    int row_tiles_max = 0, row_block = WOL_ROWS_BLOCK_A;
    int max_sx = 1, max_ry = 1, max_sz = 1, rmax_sx = 1, rmax_ry = 1;
};

// rows_block: k_search_rows instantiation to size the row tiles for (512 or 768); 0 = whichever fills its lanes better
void build_tile_tables(const int nc[3], int n, int rows_block, TileTables* t) {
    const int ncell = nc[0] * nc[1] * nc[2];
    const double wpc = std::fmax((double)n / (double)ncell, 0.05);
    // Tile shapes for WOL_NLEV local densities (waters per cell), from the box average upwards; the
    // device picks the level after the sort (k_pick_level).  A tile = sx x ry x sz cells: as many
    // waters as fit one sweep of the block (one lane per water) while the halo'd tile ((sz+2) layers
    // of (ry+2) x (sx+2) cells, each padded to an even count, +15 %) fits the LDS pair buffer and the
    // row tables; among the fullest shapes the one with the smallest halo.
    // densest tile-sized window / box average: levels in steps of ~11 % up to twice the mean, coarser beyond
    static const double lev_mult[WOL_NLEV] = {1.10, 1.22, 1.35, 1.50, 1.68, 1.90, 2.2, 2.6, 3.2, 4.0, 5.5, 8.0};
    t->nlev = WOL_NLEV;
    t->ntiles = 0;
    t->max_sx = t->max_ry = t->max_sz = 1;
    for (int l = 0; l < WOL_NLEV; ++l) {
        // a level is picked while the densest window stays below lev_wpc; its tiles are sized for the
        // typical tile of such a frame (the densest windows of a homogeneous liquid sit ~10 % above
        // the mean; a tile that does get more waters than lanes takes a second sweep)
        t->lev_wpc[l] = (float)(wpc * lev_mult[l]);
        const double w = wpc * lev_mult[l] / 1.10;
        double best = -1.0;
        int bsx = 1, bry = 1, bsz = 1, bntx = nc[0], bnty = nc[1], bntz = nc[2];
        for (int sx0 = 1; sx0 <= std::min(nc[0], WOL_FAST_MAXSX); ++sx0) {
            const int ntx = (nc[0] + sx0 - 1) / sx0, sx = (nc[0] + ntx - 1) / ntx;  // equalised
            if (sx != sx0) continue;
            for (int sz0 = 1; sz0 <= std::min(nc[2], WOL_FAST_MAXSZ); ++sz0) {
                const int ntz = (nc[2] + sz0 - 1) / sz0, sz = (nc[2] + ntz - 1) / ntz;
                if (sz != sz0) continue;
                for (int ry0 = 1; ry0 <= std::min(nc[1], WOL_FAST_MAXRY); ++ry0) {
                    const int nty = (nc[1] + ry0 - 1) / ry0, ry = (nc[1] + nty - 1) / nty;
                    if (ry != ry0) continue;
                    if (sz * ry > WOL_FAST_MAXQROWS || (sz + 2) * (ry + 2) > WOL_FAST_MAXROWS) continue;
                    const double q = sx * ry * sz * w;
                    const bool unit = (sx == 1 && ry == 1 && sz == 1);
                    if (q > (double)WOL_FAST_TARGET && !unit) continue;
                    const double staged = (double)(sz + 2) * (ry + 2) * (sx + 2) * (w + 0.5);
                    if (staged * 1.15 > 2.0 * WOL_FAST_CAPP && !unit) continue;
                    // fill of the block's lanes first (5 % bands), then the halo ratio
                    const double band = std::floor(q / (0.05 * WOL_FAST_TARGET));
                    const double score = band - staged / (q * 100.0);
                    if (score > best) { best = score; bsx = sx; bry = ry; bsz = sz; bntx = ntx; bnty = nty; bntz = ntz; }
                }
            }
        }
        t->lev_sx[l] = (short)bsx; t->lev_ry[l] = (short)bry; t->lev_sz[l] = (short)bsz;
        t->lev_ntx[l] = (short)bntx; t->lev_nty[l] = (short)bnty; t->lev_ntz[l] = (short)bntz;
        t->lev_ntiles[l] = bntx * bnty * bntz;
        t->ntiles = std::max(t->ntiles, t->lev_ntiles[l]);
        t->max_sx = std::max(t->max_sx, bsx); t->max_ry = std::max(t->max_ry, bry); t->max_sz = std::max(t->max_sz, bsz);
    }
    // ---- tiles of the row-merged kernel (wol_search_rows.hip): sx cells x ry rows in one z layer;
    //      staged = (ry + 2) triples of three z rows, sx + 2 cells long ----
    // the tile search for one instantiation (block threads, capp staged pairs); returns the lane fill at level 0
    struct RowTiles { short sx[WOL_NLEV], ry[WOL_NLEV], ntx[WOL_NLEV], nty[WOL_NLEV]; double fill0; };
    // What a tile costs the proposal kernel, fitted to the kernel's time over forced shapes on the 100 000-water box and the
    // config-5 slab (tools/tile_sweep.sh, profiles/r03_tile_sweep.txt; 3.5 % rms over 26 shapes), in units of ~1 ns of a
    // block slot: a fixed part per tile (barriers, tile set-up, the per-wave prologue and epilogue of the sweep), a part per
    // triple (its bin tables), per 64 own waters (one wave's sweep) and per staged water (placement).  The fixed part is a
    // quarter of a 32 x 5 tile -- which is why large tiles win even when their last sweep round is nearly empty, and why
    // the lane-fill score used before picked 8 x 18-cell tiles on slabs where 27 x 6 takes 23 % less time.
    auto tile_cost = [](int sxe, int re, double w) -> double {
        const double own = (double)sxe * re * w, staged = (double)(re + 2) * (sxe + 2) * w;
        return 6.43 + 0.49 * (re + 2) + 0.01494 * 64.0 * std::ceil(own / 64.0) + 0.01195 * staged;
    };
    auto row_tiles = [&](int block, int capp, RowTiles* rt) {
        const int target = WOL_ROWS_TARGET(block);
        for (int l = 0; l < WOL_NLEV; ++l) {
            const double w = wpc * lev_mult[l] / 1.10;
            double best = -1.0, bq = 0.0;
            int bsx = 1, bry = 1, bntx = nc[0], bnty = nc[1];
            for (int sx = 1; sx <= std::min(nc[0], WOL_ROWS_MAXSX); ++sx) {
                const int ntx = (nc[0] + sx - 1) / sx, sx_last = nc[0] - (ntx - 1) * sx;
                for (int ry = 1; ry <= std::min(nc[1], WOL_ROWS_MAXRY); ++ry) {
                    const int nty = (nc[1] + ry - 1) / ry, ry_last = nc[1] - (nty - 1) * ry;
                    const double q = sx * ry * w;
                    const bool unit = (sx == 1 && ry == 1);
                    // own waters: up to an eighth beyond the block's lanes (the sweep then takes a second, short round)
                    if (q > 1.12 * (double)target && !unit) continue;
                    const double staged = 3.0 * (ry + 2) * (sx + 2) * w;
                    if (staged * 1.12 + (ry + 2) > 2.0 * capp && !unit) continue;
                    if ((ry + 2) * (((sx + 2) * WOL_XSUB + 2) / 2) > WOL_ROWS_BINW) continue;
                    // a layer's tiles: (ntx - 1) x (nty - 1) whole ones and the shorter ones along its edges
                    const double layer = (ntx - 1) * ((nty - 1) * tile_cost(sx, ry, w) + tile_cost(sx, ry_last, w)) +
                                         (nty - 1) * tile_cost(sx_last, ry, w) + tile_cost(sx_last, ry_last, w);
                    const double score = 1.0e6 / layer;
                    if (score > best) { best = score; bsx = sx; bry = ry; bntx = ntx; bnty = nty; bq = q; }
                }
            }
            if (const char* force = getenv("WOL_FORCE_ROW_TILE")) {   // tuning hook (tools/tile_sweep.sh): "sx,ry" for every level
                int fsx = 0, fry = 0;
                if (sscanf(force, "%d,%d", &fsx, &fry) == 2 && fsx >= 1 && fsx <= std::min(nc[0], WOL_ROWS_MAXSX) && fry >= 1 &&
                    fry <= std::min(nc[1], WOL_ROWS_MAXRY) && (fry + 2) * (((fsx + 2) * WOL_XSUB + 2) / 2) <= WOL_ROWS_BINW) {
                    bsx = fsx; bry = fry; bntx = (nc[0] + fsx - 1) / fsx; bnty = (nc[1] + fry - 1) / fry; bq = fsx * fry * w;
                }
            }
            rt->sx[l] = (short)bsx; rt->ry[l] = (short)bry; rt->ntx[l] = (short)bntx; rt->nty[l] = (short)bnty;
            if (l == 0) rt->fill0 = std::fmin(bq / (double)block, 1.0);
        }
    };
    RowTiles rta, rtb;
    row_tiles(WOL_ROWS_BLOCK_A, WOL_ROWS_CAPP_A, &rta);
    row_tiles(WOL_ROWS_BLOCK_B, WOL_ROWS_CAPP_B, &rtb);
    // three 512-thread blocks per CU overlap their staging phases better than two of 768: the larger block has to
    // win clearly on lane fill
    const int pick = rows_block ? rows_block : (rtb.fill0 > rta.fill0 + 0.06 ? WOL_ROWS_BLOCK_B : WOL_ROWS_BLOCK_A);
    const RowTiles& rt = pick == WOL_ROWS_BLOCK_B ? rtb : rta;
    t->row_block = pick;
    t->row_tiles_max = 0;
    t->rmax_sx = t->rmax_ry = 1;
    for (int l = 0; l < WOL_NLEV; ++l) {
        t->row_sx[l] = rt.sx[l]; t->row_ry[l] = rt.ry[l]; t->row_ntx[l] = rt.ntx[l]; t->row_nty[l] = rt.nty[l];
        t->row_ntiles[l] = rt.ntx[l] * rt.nty[l] * nc[2];
        t->row_tiles_max = std::max(t->row_tiles_max, t->row_ntiles[l]);
        t->rmax_sx = std::max(t->rmax_sx, (int)rt.sx[l]);
        t->rmax_ry = std::max(t->rmax_ry, (int)rt.ry[l]);
    }
}

// the tables of (cell counts, waters, requested instantiation), memoised per host thread
const TileTables& tile_tables(const int nc[3], int n, int rows_block) {
    struct Entry { int nc[3], n, rb; TileTables t; };
    static thread_local std::vector<Entry> cache;
    for (size_t i = 0; i < cache.size(); ++i) {
        const Entry& e = cache[i];
        if (e.nc[0] == nc[0] && e.nc[1] == nc[1] && e.nc[2] == nc[2] && e.n == n && e.rb == rows_block) {
            if (i != 0) std::swap(cache[0], cache[i]);   // most recently used first
            return cache[0].t;
        }
    }
    if (cache.size() >= 32) cache.pop_back();
    Entry e;
    e.nc[0] = nc[0]; e.nc[1] = nc[1]; e.nc[2] = nc[2]; e.n = n; e.rb = rows_block;
    build_tile_tables(nc, n, rows_block, &e.t);
    cache.insert(cache.begin(), e);
    return cache[0].t;
}

// Builds the uniform grid of one frame.  Returns false when the frame must take the all-pairs
// path: a non-periodic axis (BoxL < 0), BoxL == 0 or a non-finite length.
// rows_block: k_search_rows instantiation to size the row tiles for (512 or 768); 0 = whichever fills its lanes better
bool make_grid(const double* box, int n, double cell_width, double high, FrameGrid* g, int rows_block = 0) {
    g->box = make_box(box);
    g->maxshell = -1;
    g->ncell = 1;
    for (int k = 0; k < 3; ++k) { g->nc[k] = 1; g->cw[k] = 1.0; g->lo[k] = g->hi[k] = 0; }
    for (int k = 0; k < 3; ++k)
        if (!(box[k] > 0.0) || !std::isfinite(box[k])) return false;
    if (!(high > 0.0) || !std::isfinite(high)) return false;
    double w = cell_width > 0.5 ? cell_width : 0.5;
    const double cap = std::fmax(4.0 * (double)n, 4096.0);
    for (int iter = 0; iter < 64; ++iter) {
        double prod = 1.0;
        for (int k = 0; k < 3; ++k) {
            double c = std::floor(box[k] / w);
            if (c < 1.0) c = 1.0;
            if (c > 1024.0) c = 1024.0;
            g->nc[k] = (int)c;
            prod *= c;
        }
        if (prod <= cap) break;
        w *= 1.2;
    }
    g->ncell = g->nc[0] * g->nc[1] * g->nc[2];
    int ms = 0;
    for (int k = 0; k < 3; ++k) {
        g->cw[k] = box[k] / g->nc[k];
        double mk = std::ceil((high + 1e-6) / g->cw[k]);
        if (mk > 1e6) mk = 1e6;
        int m = (int)mk;
        if (2 * m + 1 >= g->nc[k]) { g->lo[k] = (g->nc[k] - 1) / 2; g->hi[k] = g->nc[k] / 2; }
        else { g->lo[k] = g->hi[k] = m; }
        ms = std::max(ms, std::max(g->lo[k], g->hi[k]));
    }
    g->maxshell = ms;

    // ---- tiling for the filtered kernels (wol_search_fast.hip, wol_search_rows.hip) ----
    g->fast_ok = 0;
    g->nlev = 0;
    g->ntiles = 0;
    g->scale = 1.f; g->tmax = 0.f; g->S = 1.0; g->eps_units = 0;
    g->row_tiles_max = 0; g->row_eps_units = 0; g->row_block = WOL_ROWS_BLOCK_A;
    if (g->nc[0] >= 3 && g->nc[1] >= 3 && g->nc[2] >= 3) {
        const TileTables& t = tile_tables(g->nc, n, rows_block);
        g->nlev = t.nlev;
        g->ntiles = t.ntiles;
        g->row_tiles_max = t.row_tiles_max;
        g->row_block = t.row_block;
        for (int l = 0; l < WOL_NLEV; ++l) {
            g->lev_wpc[l] = t.lev_wpc[l];
            g->lev_sx[l] = t.lev_sx[l]; g->lev_ry[l] = t.lev_ry[l]; g->lev_sz[l] = t.lev_sz[l];
            g->lev_ntx[l] = t.lev_ntx[l]; g->lev_nty[l] = t.lev_nty[l]; g->lev_ntz[l] = t.lev_ntz[l];
            g->lev_ntiles[l] = t.lev_ntiles[l];
            g->row_sx[l] = t.row_sx[l]; g->row_ry[l] = t.row_ry[l]; g->row_ntx[l] = t.row_ntx[l]; g->row_nty[l] = t.row_nty[l];
            g->row_ntiles[l] = t.row_ntiles[l];
        }
        const double cwmax = std::fmax(g->cw[0], std::fmax(g->cw[1], g->cw[2]));
        const double rmax = 1.5 * cwmax + 0.6;            // beyond the largest guaranteed radius
        const double kunits = std::ldexp(1.0, 32 - WOL_ID_BITS);
        g->S = kunits / (rmax * rmax);                      // key units over [0, rmax^2)
        g->scale = (float)std::sqrt(g->S);
        g->S = (double)g->scale * (double)g->scale;
        g->tmax = 8388608.0f + (float)(kunits - 1.0);
        // fp32 error of a key: staged coordinates are rounded once (half an ulp at the largest local
        // magnitude M), differences of two such numbers are exact, the three fmas round to integers
        // (staged coordinates are cell-local values plus a cell offset, centred on the tile: M is half
        // the halo'd tile extent)
        const double ext = std::fmax(0.5 * (t.max_sx + 2) * g->cw[0], std::fmax(0.5 * (t.max_ry + 2) * g->cw[1], 0.5 * (t.max_sz + 2) * g->cw[2]))
                           + std::fmax(g->cw[0], std::fmax(g->cw[1], g->cw[2]));
        const double M = ext * g->scale;
        const double ulpM = std::ldexp(1.0, (int)std::ceil(std::log2(M)) - 24);
        // two roundings per staged coordinate (cell-local value, then + offset), two coordinates per difference
        const double eps = 2.0 * 1.7320508 * (rmax * g->scale) * (2.0 * ulpM) + 1.5;
        g->eps_units = (int)std::ceil(eps * 1.25) + 2;
        g->fast_ok = (high >= rmax) ? 1 : 0;
        {
            // per-axis coordinate magnitudes: x up to half the staged tile length, y half the staged rows
            // (both + one cell-local value), z one and a half cells; errors add in quadrature over the axes
            auto ulp_of = [&](double ext_) {
                const double Mk = ext_ * g->scale;
                return std::ldexp(1.0, (int)std::ceil(std::log2(Mk)) - 24);
            };
            const double ux = ulp_of(0.5 * (t.rmax_sx + 2) * g->cw[0] + g->cw[0]);
            const double uy = ulp_of(0.5 * (t.rmax_ry + 2) * g->cw[1] + g->cw[1]);
            const double uz = ulp_of(2.0 * g->cw[2]);
            // |key - exact d2| <= 2 d (2 |ulp|) + 1.5 in key units (d in sqrt(key units)), padded by 25 % + 2: a slope and
            // an offset for the kernel's near-tie test at the distance in question, the value at rmax for everything else
            const double slope = 2.0 * 2.0 * std::sqrt(ux * ux + uy * uy + uz * uz);
            g->row_eps_a = (float)(1.25 * slope);
            g->row_eps_b = (float)(1.25 * 1.5 + 2.0);
            // (+ 1: the row kernel's keys are d2 (1 - 2^-20), so that the clamped maximum lands on the saturated key)
            g->row_eps_units = (int)std::ceil((slope * (rmax * g->scale) + 1.5) * 1.25) + 3;
        }
    }
    return true;
}

// grid of a persistent (grid-stride) kernel: one round of resident blocks (blocks/CU from the
// occupancy query).  None of these kernels synchronises across blocks, so a block that turns out
// not to be co-resident merely starts late.
template <typename K>
int resident_grid(wol_ctx* c, K kernel, int block, size_t dyn_lds, int64_t work_items, int* grid, int margin = 0) {
    int per_cu = 0;
    HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, block, dyn_lds));
    per_cu -= margin;
    if (per_cu < 1) per_cu = 1;
    int64_t g = (int64_t)per_cu * c->num_cus;
    if (g > work_items) g = work_items;
    if (g < 1) g = 1;
    *grid = (int)g;
    return WOL_OK;
}

int begin_timing(wol_ctx* c, std::pair<hipEvent_t, hipEvent_t>* ev, hipStream_t st = nullptr) {
    if (c->ev_free.empty()) {
        hipEvent_t a, b;
        HIP_TRY(hipEventCreate(&a));
        HIP_TRY(hipEventCreate(&b));
        *ev = {a, b};
    } else {
        *ev = c->ev_free.back();
        c->ev_free.pop_back();
    }
    HIP_TRY(hipEventRecord(ev->first, st ? st : c->stream));
    return WOL_OK;
}

int end_timing(wol_ctx* c, const std::pair<hipEvent_t, hipEvent_t>& ev, hipStream_t st = nullptr) {
    HIP_TRY(hipEventRecord(ev.second, st ? st : c->stream));
    c->ev_used.push_back(ev);
    return WOL_OK;
}

int collect_timing(wol_ctx* c) {
    if (c->ev_used.empty()) return WOL_OK;
    HIP_TRY(hipStreamSynchronize(c->stream));
    for (auto& ev : c->ev_used) {
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, ev.first, ev.second));
        c->ms_total += ms;
        c->launches += 1;
        c->ev_free.push_back(ev);
    }
    c->ev_used.clear();
    return WOL_OK;
}

int launch_cell_scan(wol_ctx* c, hipStream_t st, const FrameGrid* grids, int* cells, int F, int ncell_max);

struct HistArgs {
    unsigned long long* qhist = nullptr;  // device
    unsigned long long* dhist = nullptr;  // device
    int qbins = 0, dbins = 0;
    double dbinwidth = 0.0;
};

// What the sort stage of a batch leaves behind for the kernels that consume it.
struct BatchPlan {
    wol_ctx::SortSet* S = nullptr;     // the set of intermediate arrays this batch owns (null: no frame took the cell path)
    FrameGrid* gh = nullptr;           // host copy of the per-frame grids
    int n_cell_frames = 0;
    std::vector<int> brute_frames;     // frames without a grid (non-periodic axis, tiny inputs, forced): all-pairs kernels
    bool use_fast = false, rows_kernel = false, level_early = false, pipe = false;
    bool rec32 = false;                // the sorted records are float4 (float32 frames through the streaming sort: wol_rec)
    bool idx = false;                  // index records (wol_idx_key): S.sorted holds one int per water, coordinates come from the frames
    int rows_block = 0, tiles_max = 0, tile_any_stride = 0;
    float4* s32 = nullptr;
    unsigned* cnt4 = nullptr;
    hipStream_t ss = nullptr, qs = nullptr;   // the streams of the sort and of the proposal kernel
};

// Sort stage of a batch of F device-resident frames: per-frame grids (cells of `cell_width`, walks sized for `reach`),
// the cell sort in whichever form the batch shape calls for, and -- when the fp32-filtered q path will consume the batch
// (want_fast) -- the fine-bin order, the fp32 copy and the density levels it needs.  Everything is enqueued; on return
// bp->S holds cell-sorted fp64 records (sorted), cell_start (cells) and the grids on the device.
template <typename T>
int sort_batch(wol_ctx* c, const FramesView<T>& fv, int F, const double* boxes_host, double cell_width, double reach,
               bool want_fast, bool force_brute, int allow_pipe, BatchPlan* bp, bool need_set = false, bool want_rec32 = false) {
    const int n = fv.n;

    // --- per-frame grids (host) ---
    if (c->grid_copy_pending) {  // the pinned staging buffer is still being read by an earlier upload
        HIP_TRY(hipEventSynchronize(c->grid_ev));
        c->grid_copy_pending = false;
    }
    WOL_TRY(c->grids_host.ensure(sizeof(FrameGrid) * (size_t)F));
    FrameGrid* gh = c->grids_host.as<FrameGrid>();
    int64_t cell_total = 0;
    int n_cell_frames = 0;
    std::vector<int>& brute_frames = bp->brute_frames;
    brute_frames.clear();
    const double high = reach;
    int rows_block = c->force_rows_block;   // the k_search_rows instantiation of this batch: forced, or the first frame's choice
    for (int f = 0; f < F; ++f) {
        bool ok = !force_brute && make_grid(boxes_host + 3 * (size_t)f, n, cell_width, high, &gh[f], rows_block);
        if (ok && rows_block == 0) rows_block = gh[f].row_block;
        if (!ok) {
            make_grid(boxes_host + 3 * (size_t)f, n, cell_width, high, &gh[f], rows_block);
            gh[f].maxshell = -1;  // search kernel skips this frame
            gh[f].ncell = 1;
            for (int k = 0; k < 3; ++k) gh[f].nc[k] = 1;
            brute_frames.push_back(f);
        } else {
            ++n_cell_frames;
        }
        gh[f].cell_base = cell_total;
        cell_total += gh[f].ncell + 1;
    }

    if (n_cell_frames > 0 || need_set) {   // (need_set: frames without a grid still get their one-cell `sorted` copy)
        wol_ctx::SortSet& S = c->sets[c->next_set];
        c->next_set = (c->next_set + 1) % WOL_NSETS;
        // the sort goes to the second stream when this is a large device-resident batch (the trajectory
        // path): it then overlaps the previous batch's search / finish kernels on the main stream
        const bool pipe = c->overlap && allow_pipe != 0 && F >= 64;
        if (!S.sorted_ev) {
            HIP_TRY(hipEventCreateWithFlags(&S.sorted_ev, hipEventDisableTiming));
            HIP_TRY(hipEventCreateWithFlags(&S.searched_ev, hipEventDisableTiming));
            HIP_TRY(hipEventCreateWithFlags(&S.free_ev, hipEventDisableTiming));
            HIP_TRY(hipEventCreateWithFlags(&S.rows_ev, hipEventDisableTiming));
            HIP_TRY(hipEventCreateWithFlags(&S.level_ev, hipEventDisableTiming));
        }
        hipStream_t ss = c->stream, qs = c->stream;   // the streams of the sort and of the proposal kernel
        if (pipe) {
            if (!c->sort_stream) HIP_TRY(hipStreamCreateWithFlags(&c->sort_stream, hipStreamNonBlocking));
            if (!c->search_stream) HIP_TRY(hipStreamCreateWithFlags(&c->search_stream, hipStreamNonBlocking));
            if (!c->in_ev) HIP_TRY(hipEventCreateWithFlags(&c->in_ev, hipEventDisableTiming));
            ss = c->sort_stream;
            if (c->stages >= 3) qs = c->search_stream;
            if (!c->inputs_resident) {
                // the frames may be the product of work the caller enqueued earlier on its stream: the call marks that
                // point before its first batch, every piped batch of the call waits for that mark (not for each other)
                HIP_TRY(hipStreamWaitEvent(ss, c->in_ev, 0));   // (recorded once per call, before its first batch: q_traj_impl)
            }
            if (S.used) HIP_TRY(hipStreamWaitEvent(ss, S.free_ev, 0));  // last readers of this set are done
        }
        if (c->scratch_ev) HIP_TRY(hipStreamWaitEvent(ss, c->scratch_ev, 0));
        else HIP_TRY(hipEventCreateWithFlags(&c->scratch_ev, hipEventDisableTiming));
        WOL_TRY(S.grids.ensure(sizeof(FrameGrid) * (size_t)F));
        WOL_TRY(S.cells.ensure(sizeof(int) * (size_t)cell_total));
        WOL_TRY(S.sorted.ensure(sizeof(double4) * (size_t)F * n));
        WOL_TRY(c->cellid.ensure(sizeof(int) * (size_t)F * n));
        WOL_TRY(c->rank.ensure(sizeof(int) * (size_t)F * n));
        // the per-frame grids go up on a stream of their own, as soon as the set's last readers are done: the copy then
        // runs under the kernels of the batch before instead of in front of this batch's sort (17 us per step)
        if (!c->grid_stream) HIP_TRY(hipStreamCreateWithFlags(&c->grid_stream, hipStreamNonBlocking));
        if (!c->grid_ev) HIP_TRY(hipEventCreateWithFlags(&c->grid_ev, hipEventDisableTiming));
        if (S.used) HIP_TRY(hipStreamWaitEvent(c->grid_stream, S.free_ev, 0));
        if (c->shell_dirty) {   // ... and so are the kernels of a shell call that borrowed a set
            if (!c->shell_ev) HIP_TRY(hipEventCreateWithFlags(&c->shell_ev, hipEventDisableTiming));
            HIP_TRY(hipEventRecord(c->shell_ev, c->stream));
            HIP_TRY(hipStreamWaitEvent(c->grid_stream, c->shell_ev, 0));
            c->shell_dirty = false;
        }
        HIP_TRY(hipMemcpyAsync(S.grids.p, gh, sizeof(FrameGrid) * (size_t)F, hipMemcpyHostToDevice, c->grid_stream));
        // one small control block per batch: the eight tile counters of the proposal kernel (64 B apart), then the
        // length of the exact list -- zeroed with one fill, out of the kernels' way as well
        WOL_TRY(S.work.ensure(sizeof(int) * (16 * 8 + 4)));
        HIP_TRY(hipMemsetAsync(S.work.p, 0, sizeof(int) * (16 * 8 + 4), c->grid_stream));
        HIP_TRY(hipEventRecord(c->grid_ev, c->grid_stream));
        HIP_TRY(hipStreamWaitEvent(ss, c->grid_ev, 0));
        c->grid_copy_pending = true;
        bool use_fast = want_fast;
        int tiles_max = 0;
        // the row-merged proposal kernel needs the rows bin-sorted by k_sort_in_row: rows of up to WOL_L2_MAXNX cells
        bool rows_kernel = c->search_rows && n <= (1 << WOL_SLOT_BITS);
        for (int f = 0; f < F && rows_kernel; ++f)
            if (gh[f].maxshell >= 0 && gh[f].nc[0] > WOL_L2_MAXNX) rows_kernel = false;
        for (int f = 0; f < F && use_fast; ++f) {
            if (gh[f].maxshell < 0) continue;
            if (!gh[f].fast_ok) use_fast = false;
            tiles_max = std::max(tiles_max, rows_kernel ? gh[f].row_tiles_max : gh[f].ntiles);
        }
        if (tiles_max == 0) use_fast = false;
        float4* s32 = nullptr;
        if (use_fast) {
            WOL_TRY(S.sorted32.ensure(sizeof(float4) * (size_t)F * n));
            s32 = S.sorted32.as<float4>();
        }
        // sort: one block per frame with the cell counters in LDS when the batch can fill the chip that
        // way (a block moves ~24 GB/s, one CU's share, so it needs >= ~128 frames to beat the global-
        // atomic form) and every frame's table fits; else the count / scan / scatter kernels
        int ncell_max = 0;
        for (int f = 0; f < F; ++f) ncell_max = std::max(ncell_max, gh[f].ncell);
        const size_t sort_lds = sizeof(int) * ((size_t)ncell_max + 1);
        const bool lds_sort = F >= 128 && sort_lds + sizeof(int) * WOL_SORT_BLOCK + 1024 <= 160 * 1024 && c->lds_sort;
        // streaming two-level sort ((z, y) row buckets, then a counting sort inside every row): one block per frame
        // when the batch has enough frames to fill the chip that way, else every frame split over P blocks
        // (count / scan / scatter launches) -- big frames come in small batches
        bool layer_sort = c->lds_sort && (F >= 64 || (int64_t)F * n >= 400000);
        int nz_max = 0, nrow_max = 0;
        for (int f = 0; f < F && layer_sort; ++f) {
            if (gh[f].maxshell < 0) { layer_sort = false; break; }
            if (gh[f].nc[0] > WOL_L2_MAXNX || gh[f].nc[1] * gh[f].nc[2] > WOL_L1_MAXROWS_BIG) layer_sort = false;
            nz_max = std::max(nz_max, gh[f].nc[2]);
            nrow_max = std::max(nrow_max, gh[f].nc[1] * gh[f].nc[2]);
        }
        const bool big_rows = nrow_max > WOL_L1_MAXROWS;
        // float32 frames on their way to the row-merged q pipeline keep their single-precision values through both sort levels
        // (16-byte records: half the sort's traffic, half the bytes of the certification kernel's gathers); every other consumer
        // of a sorted set reads double4 records
        const bool rec32 = want_rec32 && std::is_same<T, float>::value && layer_sort && use_fast && rows_kernel && !pipe && c->rec32;
        // float64 frames of bare O rows (3 doubles per water, up to 2^20 waters) in the single-pass form of the first level:
        // index records (wol_idx_key) -- decided below, once the form of the first level is known
        const bool idx_ok = want_rec32 && std::is_same<T, double>::value && n <= (1 << WOL_IDX_BITS) && layer_sort &&
                            use_fast && rows_kernel && !pipe && c->idx_sort && c->seg_sort;
        bool idx = idx_ok;   // (tentatively: cleared below when the first level takes a form other than the single-pass one)
        const size_t rec_size = (rec32 || idx_ok) ? sizeof(float4) : sizeof(double4);
        // second level of every sort: the rows (cell-sorted or merely bucketed) are sorted by fine x bin, with the bin
        // populations per cell, into the final arrays.  The row-merged proposal kernel stages from rows in that order.
        const bool in_row_stage = layer_sort || (use_fast && rows_kernel);
        int nz_all = 0, nrow_all = 0, nb_cap = WOL_XSUB;
        for (int f = 0; f < F; ++f) {
            if (gh[f].maxshell < 0) continue;
            nz_all = std::max(nz_all, gh[f].nc[2]);
            nrow_all = std::max(nrow_all, gh[f].nc[1] * gh[f].nc[2]);
            nb_cap = std::max(nb_cap, std::min(gh[f].nc[0], WOL_L2_MAXNX) * WOL_XSUB);
        }
        int ny_all_frames = 1;
        for (int f = 0; f < F; ++f) ny_all_frames = std::max(ny_all_frames, gh[f].nc[1]);
        unsigned* cnt4 = nullptr;
        if (use_fast && rows_kernel) {
            WOL_TRY(S.cnt4.ensure(sizeof(unsigned) * (size_t)cell_total));
            cnt4 = S.cnt4.as<unsigned>();
        }
        if (in_row_stage) WOL_TRY(c->sort_tmp.ensure(sizeof(double4) * (size_t)F * n));
        RowPieces pieces{nullptr, 0, 0, 0, 0};
        const bool in_row_long = (int64_t)n > 100 * (int64_t)std::max(nrow_all, 1);   // long rows (big boxes): four waters per lane
        // dynamic LDS of the second level: the waves' bin counters and, in the segment form, the layer's copy of the row table
        auto in_row_lds = [&](int nchunk, int ny_cap) -> size_t {
            return sizeof(int) * 2 * (size_t)nb_cap * (WOL_BLOCK / 64) +
                   (nchunk ? sizeof(int) * (size_t)(WOL_BLOCK / 64) * (2 * nchunk + 1) + sizeof(unsigned short) * (size_t)nchunk * ny_cap + 8 : 0);
        };
        // does it fit next to the kernel's static arrays?  (raises the kernel's dynamic-LDS limit when it has to: a long thin
        // box has hundreds of rows per layer and a hundred chunks)
        auto in_row_fits = [&](size_t dyn) -> bool {
            const void* fn = idx ? (in_row_long ? reinterpret_cast<const void*>(&k_sort_in_row<4, float4, true>) : reinterpret_cast<const void*>(&k_sort_in_row<2, float4, true>))
                           : rec32 ? (in_row_long ? reinterpret_cast<const void*>(&k_sort_in_row<4, float4>) : reinterpret_cast<const void*>(&k_sort_in_row<2, float4>))
                                   : (in_row_long ? reinterpret_cast<const void*>(&k_sort_in_row<4>) : reinterpret_cast<const void*>(&k_sort_in_row<2>));
            size_t& stat = c->in_row_static_lds[(in_row_long ? 1 : 0) + (idx ? 4 : rec32 ? 2 : 0)];
            size_t& limit = c->in_row_dyn_limit[(in_row_long ? 1 : 0) + (idx ? 4 : rec32 ? 2 : 0)];
            if (!stat) {
                hipFuncAttributes fa_;
                if (hipFuncGetAttributes(&fa_, fn) != hipSuccess) return false;
                stat = fa_.sharedSizeBytes ? fa_.sharedSizeBytes : 1;
                limit = 64 * 1024 > stat ? 64 * 1024 - stat : 0;
            }
            if (stat + dyn > 160 * 1024) return false;
            if (dyn > limit) {
                if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(160 * 1024 - stat)) != hipSuccess) {
                    (void)hipGetLastError();
                    return false;
                }
                limit = 160 * 1024 - stat;
            }
            return true;
        };
        auto launch_in_row = [&](int ro_stride) {
            const size_t lds = in_row_lds(pieces.tab ? pieces.nchunk_cap : 0, pieces.ny_cap);
            if (idx && in_row_long)
                hipLaunchKernelGGL((k_sort_in_row<4, float4, true>), dim3(std::max(nz_all, 1), F), dim3(WOL_BLOCK), lds, ss, S.grids.as<FrameGrid>(),
                                   c->layer_off.as<int>(), ro_stride, c->sort_tmp.as<float4>(), n, S.cells.as<int>(),
                                   S.sorted.as<int>(), s32, cnt4, nb_cap, pieces);
            else if (idx)
                hipLaunchKernelGGL((k_sort_in_row<2, float4, true>), dim3(std::max(nz_all, 1), F), dim3(WOL_BLOCK), lds, ss, S.grids.as<FrameGrid>(),
                                   c->layer_off.as<int>(), ro_stride, c->sort_tmp.as<float4>(), n, S.cells.as<int>(),
                                   S.sorted.as<int>(), s32, cnt4, nb_cap, pieces);
            else if (rec32 && in_row_long)
                hipLaunchKernelGGL((k_sort_in_row<4, float4>), dim3(std::max(nz_all, 1), F), dim3(WOL_BLOCK), lds, ss, S.grids.as<FrameGrid>(),
                                   c->layer_off.as<int>(), ro_stride, c->sort_tmp.as<float4>(), n, S.cells.as<int>(),
                                   S.sorted.as<float4>(), s32, cnt4, nb_cap, pieces);
            else if (rec32)
                hipLaunchKernelGGL((k_sort_in_row<2, float4>), dim3(std::max(nz_all, 1), F), dim3(WOL_BLOCK), lds, ss, S.grids.as<FrameGrid>(),
                                   c->layer_off.as<int>(), ro_stride, c->sort_tmp.as<float4>(), n, S.cells.as<int>(),
                                   S.sorted.as<float4>(), s32, cnt4, nb_cap, pieces);
            else if (in_row_long)
                hipLaunchKernelGGL(k_sort_in_row<4>, dim3(std::max(nz_all, 1), F), dim3(WOL_BLOCK), lds, ss, S.grids.as<FrameGrid>(),
                                   c->layer_off.as<int>(), ro_stride, c->sort_tmp.as<double4>(), n, S.cells.as<int>(),
                                   S.sorted.as<double4>(), s32, cnt4, nb_cap, pieces);
            else
                hipLaunchKernelGGL(k_sort_in_row<2>, dim3(std::max(nz_all, 1), F), dim3(WOL_BLOCK), lds, ss, S.grids.as<FrameGrid>(),
                                   c->layer_off.as<int>(), ro_stride, c->sort_tmp.as<double4>(), n, S.cells.as<int>(),
                                   S.sorted.as<double4>(), s32, cnt4, nb_cap, pieces);
        };
        // (the cell-level sorts below write into the row buckets when the in-row stage follows them)
        double4* cell_sorted_out = (in_row_stage && !layer_sort) ? c->sort_tmp.as<double4>() : S.sorted.as<double4>();
        float4* cell_sorted_s32 = (in_row_stage && !layer_sort) ? nullptr : s32;
        auto finish_cell_sort = [&]() -> int {
            if (!in_row_stage || layer_sort) return WOL_OK;
            const int ro_stride = nrow_all + 1;
            WOL_TRY(c->layer_off.ensure(sizeof(int) * (size_t)F * ro_stride));
            hipLaunchKernelGGL(k_rows_from_cells, dim3((nrow_all + WOL_BLOCK) / WOL_BLOCK, F), dim3(WOL_BLOCK), 0, ss,
                               S.grids.as<FrameGrid>(), S.cells.as<int>(), c->layer_off.as<int>(), ro_stride);
            launch_in_row(ro_stride);
            return WOL_OK;
        };
        int P = 1;
        bool level_early = false;
        if (layer_sort && (5 * F < 3 * c->num_cus || big_rows)) {   // fewer frames than ~0.6 blocks per CU: split them
            P = std::min(64, std::max(1, (2 * c->num_cus + F - 1) / F));          // about two blocks per CU
            P = std::min(P, std::max(1, n / (4 * WOL_SORT_BLOCK)));                  // at least a few chunks each
            if (P == 1 && big_rows) P = 2;                                           // the fused kernel's tables end at MAXROWS
        }
        if (layer_sort && c->sort_parts > 0) P = std::max(c->sort_parts, big_rows ? 2 : 1);   // tuning override (big row tables need the split form)
        if (layer_sort) {
            const int ro_stride = nrow_max + 1;
            WOL_TRY(c->layer_off.ensure(sizeof(int) * (size_t)F * ro_stride));
            if (!c->sort_rows_lds_set) {
                const int lds_a = (int)wol_sort_rows_lds(WOL_L1_MAXROWS, WOL_L1_CHUNK);
                const int lds_b = (int)wol_sort_rows_lds(WOL_L1_MAXROWS_BIG, WOL_L1_CHUNK_BIG);
                const int lds_p = (int)wol_sort_rows_lds(WOL_L1_MAXROWS, WOL_L1_CHUNK_PIPE, WOL_L1_BLOCK_PIPE);
                HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_sort_rows_seg<double, WOL_SORT_BLOCK, WOL_L1_CHUNK>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_a));
                HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_sort_rows_seg<float, WOL_SORT_BLOCK, WOL_L1_CHUNK>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_a));
                HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_sort_rows<double, WOL_SORT_BLOCK, WOL_L1_CHUNK>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_a));
                HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_sort_rows<float, WOL_SORT_BLOCK, WOL_L1_CHUNK>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_a));
                HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_sort_rows<double, WOL_L1_BLOCK_PIPE, WOL_L1_CHUNK_PIPE>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_p));
                HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_sort_rows<float, WOL_L1_BLOCK_PIPE, WOL_L1_CHUNK_PIPE>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_p));
                HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_sort_rows_scatter<double, WOL_L1_CHUNK>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_a));
                HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_sort_rows_scatter<float, WOL_L1_CHUNK>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_a));
                HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_sort_rows_scatter<double, WOL_L1_CHUNK_BIG>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_b));
                HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_sort_rows_scatter<float, WOL_L1_CHUNK_BIG>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_b));
                HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_sort_rows_seg<float, WOL_SORT_BLOCK, WOL_L1_CHUNK, float4>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_a));
                HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_sort_rows_seg<double, WOL_SORT_BLOCK, WOL_L1_CHUNK, float4, true>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_a));
                HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_sort_rows_seg<double, WOL_SORT_BLOCK, WOL_L1_CHUNK_BIG, float4, true>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_b));
                HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_sort_rows_seg<float, WOL_SORT_BLOCK, WOL_L1_CHUNK_BIG, float4>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_b));
                HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_sort_rows_seg<float, WOL_SORT_BLOCK, WOL_L1_CHUNK_BIG>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_b));
                HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_sort_rows_seg<double, WOL_SORT_BLOCK, WOL_L1_CHUNK_BIG>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_b));
                HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_sort_rows<float, WOL_SORT_BLOCK, WOL_L1_CHUNK, float4>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_a));
                HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_sort_rows_scatter<float, WOL_L1_CHUNK, float4>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_a));
                HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_sort_rows_scatter<float, WOL_L1_CHUNK_BIG, float4>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_b));
                // (16-byte records: chunks twice as long fit beside the big row tables)
                HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_sort_rows_scatter<float, 2 * WOL_L1_CHUNK_BIG, float4>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                            (int)wol_sort_rows_lds(WOL_L1_MAXROWS_BIG, 2 * WOL_L1_CHUNK_BIG, WOL_SORT_BLOCK, sizeof(float4))));
                c->sort_rows_lds_set = 1;
            }
            // (chunks of the split single-pass form below: smaller where the row tables are big)
            const int ch_p = big_rows ? WOL_L1_CHUNK_BIG : WOL_L1_CHUNK;
            const int nchunk_p = (n + ch_p * WOL_SORT_BLOCK - 1) / (ch_p * WOL_SORT_BLOCK);
            if (P == 1 && pipe) {
                idx = false;
                // beside the proposal kernel of the previous batch: eight waves and a third of the LDS per block, so that
                // the two share a CU instead of taking turns
                hipLaunchKernelGGL((k_sort_rows<T, WOL_L1_BLOCK_PIPE, WOL_L1_CHUNK_PIPE>), dim3(F), dim3(WOL_L1_BLOCK_PIPE),
                                   wol_sort_rows_lds(nrow_max, WOL_L1_CHUNK_PIPE, WOL_L1_BLOCK_PIPE), ss, fv,
                                   S.grids.as<FrameGrid>(), c->layer_off.as<int>(), ro_stride, c->sort_tmp.as<double4>(), nrow_max);
            } else if (P == 1 && c->seg_sort && (n + WOL_L1_CHUNK * WOL_SORT_BLOCK - 1) / (WOL_L1_CHUNK * WOL_SORT_BLOCK) <= WOL_SEG_MAXCHUNK &&
                       (c->seg_sort >= 2 || n >= 8 * WOL_L1_CHUNK * WOL_SORT_BLOCK) &&
                       in_row_fits(in_row_lds((n + WOL_L1_CHUNK * WOL_SORT_BLOCK - 1) / (WOL_L1_CHUNK * WOL_SORT_BLOCK), ny_all_frames + 1))) {
                // single pass: chunks written as they lie in LDS plus a table of the rows' starts inside every chunk
                const int chunk_waters = WOL_L1_CHUNK * WOL_SORT_BLOCK, nchunk = (n + chunk_waters - 1) / chunk_waters;
                WOL_TRY(c->seg_tab.ensure(sizeof(unsigned short) * (size_t)F * (size_t)(nrow_max + 1) * nchunk));
                pieces = RowPieces{c->seg_tab.as<unsigned short>(), nrow_max + 1, nchunk, chunk_waters, ny_all_frames + 1};
                if constexpr (std::is_same<T, double>::value) {
                    if (idx)
                        hipLaunchKernelGGL((k_sort_rows_seg<double, WOL_SORT_BLOCK, WOL_L1_CHUNK, float4, true>), dim3(1, F), dim3(WOL_SORT_BLOCK),
                                           wol_sort_rows_lds(nrow_max, WOL_L1_CHUNK, WOL_SORT_BLOCK, rec_size), ss, fv,
                                           S.grids.as<FrameGrid>(), c->layer_off.as<int>(), ro_stride, c->sort_tmp.as<float4>(),
                                           c->seg_tab.as<unsigned short>(), nrow_max + 1, nchunk, nrow_max, 1, (int*)nullptr, 0);
                }
                if constexpr (std::is_same<T, float>::value) {
                    if (rec32)
                        hipLaunchKernelGGL((k_sort_rows_seg<float, WOL_SORT_BLOCK, WOL_L1_CHUNK, float4>), dim3(1, F), dim3(WOL_SORT_BLOCK),
                                           wol_sort_rows_lds(nrow_max, WOL_L1_CHUNK, WOL_SORT_BLOCK, rec_size), ss, fv,
                                           S.grids.as<FrameGrid>(), c->layer_off.as<int>(), ro_stride, c->sort_tmp.as<float4>(),
                                           c->seg_tab.as<unsigned short>(), nrow_max + 1, nchunk, nrow_max, 1, (int*)nullptr, 0);
                }
                if (!rec32 && !idx)
                hipLaunchKernelGGL((k_sort_rows_seg<T, WOL_SORT_BLOCK, WOL_L1_CHUNK>), dim3(1, F), dim3(WOL_SORT_BLOCK), wol_sort_rows_lds(nrow_max, WOL_L1_CHUNK), ss, fv,
                                   S.grids.as<FrameGrid>(), c->layer_off.as<int>(), ro_stride, c->sort_tmp.as<double4>(),
                                   c->seg_tab.as<unsigned short>(), nrow_max + 1, nchunk, nrow_max, 1, (int*)nullptr, 0);
            } else if (P == 1) {
                idx = false;
                if constexpr (std::is_same<T, float>::value) {
                    if (rec32)
                        hipLaunchKernelGGL((k_sort_rows<float, WOL_SORT_BLOCK, WOL_L1_CHUNK, float4>), dim3(F), dim3(WOL_SORT_BLOCK),
                                           wol_sort_rows_lds(nrow_max, WOL_L1_CHUNK, WOL_SORT_BLOCK, rec_size), ss, fv,
                                           S.grids.as<FrameGrid>(), c->layer_off.as<int>(), ro_stride, c->sort_tmp.as<float4>(), nrow_max);
                }
                if (!rec32)
                hipLaunchKernelGGL((k_sort_rows<T, WOL_SORT_BLOCK, WOL_L1_CHUNK>), dim3(F), dim3(WOL_SORT_BLOCK), wol_sort_rows_lds(nrow_max, WOL_L1_CHUNK), ss, fv,
                                   S.grids.as<FrameGrid>(), c->layer_off.as<int>(), ro_stride, c->sort_tmp.as<double4>(), nrow_max);
            } else if (c->seg_sort && !pipe && nchunk_p <= WOL_SEG_MAXCHUNK && 10 * ch_p * WOL_SORT_BLOCK >= 13 * nrow_max &&
                       in_row_fits(in_row_lds(nchunk_p, ny_all_frames + 1))) {
                // (a chunk must hold more than a water or so of every row: the second level gathers a row from its pieces in all
                //  the chunks -- on the config-5 slab, 2 048-water chunks over 4 428 rows, it took 462 us instead of 219)
                // big frames in small batches, single pass: P blocks per frame write the frame's chunks as they lie in LDS (block p
                // the chunks p, p + P, ...) with the table of the rows' starts; their row totals are added up and scanned by
                // k_sort_rows_scan -- no counting pass over the frame, one contiguous write stream per block
                const int ch = ch_p, chunk_waters = ch * WOL_SORT_BLOCK, nchunk = nchunk_p;
                P = std::min(P, nchunk);
                WOL_TRY(c->seg_tab.ensure(sizeof(unsigned short) * (size_t)F * (size_t)(nrow_max + 1) * nchunk));
                WOL_TRY(c->rank.ensure(sizeof(int) * (size_t)F * P * ro_stride));   // per-block row totals
                int* cntb = c->rank.as<int>();
                pieces = RowPieces{c->seg_tab.as<unsigned short>(), nrow_max + 1, nchunk, chunk_waters, ny_all_frames + 1};
                const size_t lds1 = wol_sort_rows_lds(nrow_max, ch, WOL_SORT_BLOCK, rec_size);
                bool done = false;
                if constexpr (std::is_same<T, double>::value) {
                    if (idx) {
                        if (big_rows)
                            hipLaunchKernelGGL((k_sort_rows_seg<double, WOL_SORT_BLOCK, WOL_L1_CHUNK_BIG, float4, true>), dim3(P, F), dim3(WOL_SORT_BLOCK), lds1, ss, fv,
                                               S.grids.as<FrameGrid>(), c->layer_off.as<int>(), ro_stride, c->sort_tmp.as<float4>(),
                                               c->seg_tab.as<unsigned short>(), nrow_max + 1, nchunk, nrow_max, P, cntb, ro_stride);
                        else
                            hipLaunchKernelGGL((k_sort_rows_seg<double, WOL_SORT_BLOCK, WOL_L1_CHUNK, float4, true>), dim3(P, F), dim3(WOL_SORT_BLOCK), lds1, ss, fv,
                                               S.grids.as<FrameGrid>(), c->layer_off.as<int>(), ro_stride, c->sort_tmp.as<float4>(),
                                               c->seg_tab.as<unsigned short>(), nrow_max + 1, nchunk, nrow_max, P, cntb, ro_stride);
                        done = true;
                    }
                }
                if constexpr (std::is_same<T, float>::value) {
                    if (rec32) {
                        if (big_rows)
                            hipLaunchKernelGGL((k_sort_rows_seg<float, WOL_SORT_BLOCK, WOL_L1_CHUNK_BIG, float4>), dim3(P, F), dim3(WOL_SORT_BLOCK), lds1, ss, fv,
                                               S.grids.as<FrameGrid>(), c->layer_off.as<int>(), ro_stride, c->sort_tmp.as<float4>(),
                                               c->seg_tab.as<unsigned short>(), nrow_max + 1, nchunk, nrow_max, P, cntb, ro_stride);
                        else
                            hipLaunchKernelGGL((k_sort_rows_seg<float, WOL_SORT_BLOCK, WOL_L1_CHUNK, float4>), dim3(P, F), dim3(WOL_SORT_BLOCK), lds1, ss, fv,
                                               S.grids.as<FrameGrid>(), c->layer_off.as<int>(), ro_stride, c->sort_tmp.as<float4>(),
                                               c->seg_tab.as<unsigned short>(), nrow_max + 1, nchunk, nrow_max, P, cntb, ro_stride);
                        done = true;
                    }
                }
                if (!done) {
                    if (big_rows)
                        hipLaunchKernelGGL((k_sort_rows_seg<T, WOL_SORT_BLOCK, WOL_L1_CHUNK_BIG>), dim3(P, F), dim3(WOL_SORT_BLOCK), lds1, ss, fv,
                                           S.grids.as<FrameGrid>(), c->layer_off.as<int>(), ro_stride, c->sort_tmp.as<double4>(),
                                           c->seg_tab.as<unsigned short>(), nrow_max + 1, nchunk, nrow_max, P, cntb, ro_stride);
                    else
                        hipLaunchKernelGGL((k_sort_rows_seg<T, WOL_SORT_BLOCK, WOL_L1_CHUNK>), dim3(P, F), dim3(WOL_SORT_BLOCK), lds1, ss, fv,
                                           S.grids.as<FrameGrid>(), c->layer_off.as<int>(), ro_stride, c->sort_tmp.as<double4>(),
                                           c->seg_tab.as<unsigned short>(), nrow_max + 1, nchunk, nrow_max, P, cntb, ro_stride);
                }
                hipLaunchKernelGGL(k_sort_rows_scan, dim3(F), dim3(WOL_SORT_BLOCK), 0, ss, S.grids.as<FrameGrid>(), cntb, ro_stride, P,
                                   c->layer_off.as<int>(), ro_stride);
            } else {
                idx = false;
                WOL_TRY(c->rank.ensure(sizeof(int) * (size_t)F * P * ro_stride));   // per-block row counts, then cursors
                int* cntb = c->rank.as<int>();
                hipLaunchKernelGGL(k_sort_rows_count<T>, dim3(P, F), dim3(WOL_SORT_BLOCK), sizeof(int) * (size_t)(nrow_max + 1), ss, fv,
                                   S.grids.as<FrameGrid>(), cntb, ro_stride, P);
                hipLaunchKernelGGL(k_sort_rows_scan, dim3(F), dim3(WOL_SORT_BLOCK), 0, ss, S.grids.as<FrameGrid>(), cntb, ro_stride, P,
                                   c->layer_off.as<int>(), ro_stride);
                if constexpr (std::is_same<T, float>::value) {
                    if (rec32 && big_rows)
                        hipLaunchKernelGGL((k_sort_rows_scatter<float, 2 * WOL_L1_CHUNK_BIG, float4>), dim3(P, F), dim3(WOL_SORT_BLOCK),
                                           wol_sort_rows_lds(nrow_max, 2 * WOL_L1_CHUNK_BIG, WOL_SORT_BLOCK, rec_size), ss, fv, S.grids.as<FrameGrid>(), cntb, ro_stride, P,
                                           c->sort_tmp.as<float4>(), nrow_max);
                    else if (rec32)
                        hipLaunchKernelGGL((k_sort_rows_scatter<float, WOL_L1_CHUNK, float4>), dim3(P, F), dim3(WOL_SORT_BLOCK),
                                           wol_sort_rows_lds(nrow_max, WOL_L1_CHUNK, WOL_SORT_BLOCK, rec_size), ss, fv, S.grids.as<FrameGrid>(), cntb, ro_stride, P,
                                           c->sort_tmp.as<float4>(), nrow_max);
                }
                if (rec32) {
                } else if (big_rows)
                    hipLaunchKernelGGL((k_sort_rows_scatter<T, WOL_L1_CHUNK_BIG>), dim3(P, F), dim3(WOL_SORT_BLOCK),
                                       wol_sort_rows_lds(nrow_max, WOL_L1_CHUNK_BIG), ss, fv, S.grids.as<FrameGrid>(), cntb, ro_stride, P,
                                       c->sort_tmp.as<double4>(), nrow_max);
                else
                    hipLaunchKernelGGL((k_sort_rows_scatter<T, WOL_L1_CHUNK>), dim3(P, F), dim3(WOL_SORT_BLOCK),
                                       wol_sort_rows_lds(nrow_max, WOL_L1_CHUNK), ss, fv, S.grids.as<FrameGrid>(), cntb, ro_stride, P,
                                       c->sort_tmp.as<double4>(), nrow_max);
            }
            if (use_fast) {
                // the density level of every frame needs the rows' populations only: picked on the side stream while the
                // second level runs (the sort stream waits for it before the row offsets are reused)
                WOL_TRY(S.level.ensure(sizeof(int) * (size_t)F));
                HIP_TRY(hipEventRecord(S.rows_ev, ss));
                HIP_TRY(hipStreamWaitEvent(c->grid_stream, S.rows_ev, 0));
                int* tile_any = nullptr;
                if (rows_kernel) {   // which (layer, tile row) pairs hold any water: the proposal kernel skips the others
                    WOL_TRY(S.tile_any.ensure(sizeof(int) * (size_t)F * (size_t)(nrow_max + 1)));
                    tile_any = S.tile_any.as<int>();
                    bp->tile_any_stride = nrow_max + 1;
                }
                const int rows_cap = std::min(nrow_max, 12288);   // (frames with more rows keep the densest level: 48 KB of LDS at most)
                hipLaunchKernelGGL(k_pick_level, dim3(F), dim3(WOL_BLOCK), sizeof(int) * (size_t)rows_cap, c->grid_stream, S.grids.as<FrameGrid>(), S.cells.as<int>(),
                                   c->layer_off.as<int>(), ro_stride, S.level.as<int>(), tile_any, nrow_max + 1, rows_cap);
                HIP_TRY(hipEventRecord(S.level_ev, c->grid_stream));
                level_early = true;
            }
            launch_in_row(ro_stride);
            if (level_early) HIP_TRY(hipStreamWaitEvent(ss, S.level_ev, 0));
        } else if (lds_sort) {
            if (sort_lds > c->sort_lds_set) {
                HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_sort_frame_lds<T>),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)(160 * 1024 - sizeof(int) * WOL_SORT_BLOCK - 1024)));
                c->sort_lds_set = 160 * 1024;
            }
            hipLaunchKernelGGL(k_sort_frame_lds<T>, dim3(F), dim3(WOL_SORT_BLOCK), sort_lds, ss, fv,
                               S.grids.as<FrameGrid>(), S.cells.as<int>(), c->rank.as<int>(),
                               cell_sorted_out, cell_sorted_s32);
            WOL_TRY(finish_cell_sort());
        } else {
            HIP_TRY(hipMemsetAsync(S.cells.p, 0, sizeof(int) * (size_t)cell_total, ss));
            const dim3 gb((n + WOL_BLOCK - 1) / WOL_BLOCK, F);
            hipLaunchKernelGGL(k_bin_count<T>, gb, dim3(WOL_BLOCK), 0, ss, fv, S.grids.as<FrameGrid>(),
                               S.cells.as<int>(), c->cellid.as<int>(), c->rank.as<int>());
            WOL_TRY(launch_cell_scan(c, ss, S.grids.as<FrameGrid>(), S.cells.as<int>(), F, ncell_max));
            hipLaunchKernelGGL(k_scatter<T>, gb, dim3(WOL_BLOCK), 0, ss, fv, S.grids.as<FrameGrid>(),
                               S.cells.as<int>(), c->cellid.as<int>(), c->rank.as<int>(), cell_sorted_out, cell_sorted_s32);
            WOL_TRY(finish_cell_sort());
        }

        HIP_TRY(hipEventRecord(c->scratch_ev, ss));
        if (pipe) {
            HIP_TRY(hipEventRecord(S.sorted_ev, ss));
            HIP_TRY(hipStreamWaitEvent(use_fast ? qs : c->stream, S.sorted_ev, 0));
        }
        c->last_set = &S;
        bp->S = &S;
        bp->use_fast = use_fast; bp->rows_kernel = rows_kernel; bp->level_early = level_early; bp->pipe = pipe;
        bp->tiles_max = tiles_max; bp->s32 = s32; bp->cnt4 = cnt4; bp->ss = ss; bp->qs = qs; bp->rec32 = rec32; bp->idx = idx;
    }
    bp->gh = gh;
    bp->n_cell_frames = n_cell_frames;
    bp->rows_block = rows_block;
    return WOL_OK;
}

// Runs the q pipeline on F device-resident frames (views already point at device memory).
// out.* are device pointers laid out [F][ns]...; sub (device, nullable) only with F == 1.
template <typename T>
int run_q_batch(wol_ctx* c, const FramesView<T>& fv, int F, const double* boxes_host, const T* sub,
                int64_t sub_stride, int ns, double low, double high, QOut out, int allow_pipe = 0, bool need_set = false,
                bool want_rec32 = false) {
    const int n = fv.n;
    c->stat_slots = 0;   // (set again below when the certification kernel keeps per-wave partial sums)
    if (F <= 0 || ns <= 0) return WOL_OK;
    const double lowsq = low * low, highsq = high * high;
    const size_t hist_lds = (size_t)((out.qhist ? out.qbins : 0) + (out.dhist ? out.dbins : 0)) * sizeof(int);
    const bool force_brute = (c->path == 1) || (c->path == 0 && n <= 1024) || (sub != nullptr && !std::is_same<T, double>::value);
    const bool want_fast = (c->path == 0 || c->path == 3) && sub == nullptr && low == 0.0 && ns == n;
    BatchPlan bp;
    WOL_TRY(sort_batch<T>(c, fv, F, boxes_host, c->cell_width, high, want_fast, force_brute, allow_pipe, &bp, need_set, want_rec32));
    const std::vector<int>& brute_frames = bp.brute_frames;
    if (bp.n_cell_frames > 0) {
        wol_ctx::SortSet& S = *bp.S;
        const bool use_fast = bp.use_fast, rows_kernel = bp.rows_kernel, level_early = bp.level_early, pipe = bp.pipe;
        const int rows_block = bp.rows_block, tiles_max = bp.tiles_max;
        float4* s32 = bp.s32;
        unsigned* cnt4 = bp.cnt4;
        hipStream_t qs = bp.qs;

        SearchArgs a;
        a.grids = S.grids.as<FrameGrid>();
        a.sorted = S.sorted.as<double4>();
        a.cell_start = S.cells.as<int>();
        a.n = n;
        a.F = F;
        a.sub = reinterpret_cast<const double*>(sub);  // only double reaches here with sub != null
        a.sub_stride = sub_stride;
        a.ns = ns;
        a.lowsq = lowsq; a.highsq = highsq; a.high = high;
        a.out = out;
        a.tiles_per_frame = (ns + WOL_BLOCK - 1) / WOL_BLOCK;
        a.xcd_affine = (F >= 8) ? 1 : 0;
        a.sidx = nullptr; a.src = nullptr; a.src_fstride = 0; a.src_astride = 3;
        const int64_t total = (int64_t)F * a.tiles_per_frame;
        int grid = 0;
        WOL_TRY(resident_grid(c, k_search_exact, WOL_BLOCK, hist_lds, total, &grid));
        if (a.xcd_affine) grid = std::max(8, (grid / 8) * 8);
        std::pair<hipEvent_t, hipEvent_t> ev;
        if (use_fast) {
            WOL_TRY(c->slow_list.ensure(sizeof(unsigned long long) * (size_t)F * n));
            WOL_TRY(S.prop.ensure(sizeof(int4) * (size_t)F * n));
            WOL_TRY(S.prop_key.ensure(sizeof(int) * (size_t)F * n));
            // every water starts as "not certifiable": one the 3-D tile kernel does not reach (it enumerates at most 255
            // waters per cell) is then finished by the exact kernel instead of being certified from whatever the buffer
            // held.  (The row kernel writes a key for every water of every tile it is given, overflowing tiles included.)
            if (!rows_kernel) HIP_TRY(hipMemsetAsync(S.prop_key.p, 0xFF, sizeof(int) * (size_t)F * n, qs));
            else if (c->check_props) {
                HIP_TRY(hipMemsetAsync(S.prop_key.p, WOL_PROP_POISON & 0xFF, sizeof(int) * (size_t)F * n, qs));
                HIP_TRY(hipMemsetAsync(S.prop.p, 0x7F, sizeof(int4) * (size_t)F * n, qs));
            }
            if (!level_early) {
                WOL_TRY(S.level.ensure(sizeof(int) * (size_t)F));
                int nrow_all = 1;
                for (int f = 0; f < F; ++f) if (bp.gh[f].maxshell >= 0) nrow_all = std::max(nrow_all, bp.gh[f].nc[1] * bp.gh[f].nc[2]);
                const int rows_cap = std::min(nrow_all, 12288);
                hipLaunchKernelGGL(k_pick_level, dim3(F), dim3(WOL_BLOCK), sizeof(int) * (size_t)rows_cap, qs, a.grids, a.cell_start, (const int*)nullptr, 0, S.level.as<int>(),
                                   (int*)nullptr, 0, rows_cap);
            }
            FastArgs fa;
            fa.level = S.level.as<int>();
            fa.tile_any = (level_early && rows_kernel && bp.tile_any_stride && c->skip_empty) ? S.tile_any.as<int>() : nullptr;
            fa.tile_any_stride = bp.tile_any_stride;
            fa.grids = a.grids; fa.sorted = a.sorted; fa.sorted32 = s32; fa.cell_start = a.cell_start;
            fa.n = n; fa.F = F; fa.highsq = highsq; fa.high = high; fa.out = out;
            fa.tiles_per_frame = tiles_max;
            fa.xcd_affine = a.xcd_affine;
            fa.slow_list = c->slow_list.as<unsigned long long>();
            fa.slow_count = S.work.as<int>() + 16 * 8;
            fa.fast_q = (out.ang == nullptr) ? 1 : 0;
            fa.prop = S.prop.as<int4>();
            fa.prop_key = S.prop_key.as<int>();
            fa.stamps = nullptr;
            fa.rows = rows_kernel ? 1 : 0;
            fa.cnt4 = cnt4;
            fa.work = nullptr;
            if (rows_kernel && c->dyn_tiles != 0) fa.work = S.work.as<int>();
            fa.sidx = bp.idx ? S.sorted.as<int>() : nullptr;
            fa.src = reinterpret_cast<const double*>(fv.base);   // (read only with index records: T is double then)
            fa.src_fstride = fv.frame_stride; fa.src_astride = fv.atom_stride;
            a.sidx = fa.sidx; a.src = fa.src; a.src_fstride = fa.src_fstride; a.src_astride = fa.src_astride;
#ifdef WOL_PHASE_STAMPS
            WOL_TRY(c->stamps.ensure(sizeof(unsigned long long) * 8));
            fa.stamps = c->stamps.as<unsigned long long>();
#endif
            const int64_t ftotal = (int64_t)F * tiles_max;
            int fgrid = 0;
            c->last_rows_block = rows_kernel ? rows_block : 0;
            c->last_rec_bytes = bp.idx ? 4 : bp.rec32 ? 16 : 32;
            if (!rows_kernel) WOL_TRY(resident_grid(c, k_search_fast, WOL_FAST_BLOCK, 0, ftotal, &fgrid, 0));
            else if (rows_block == WOL_ROWS_BLOCK_B) WOL_TRY(resident_grid(c, k_search_rows<WOL_ROWS_BLOCK_B, WOL_ROWS_CAPP_B>, WOL_ROWS_BLOCK_B, 0, ftotal, &fgrid, c->search_margin));
            else WOL_TRY(resident_grid(c, k_search_rows<WOL_ROWS_BLOCK_A, WOL_ROWS_CAPP_A>, WOL_ROWS_BLOCK_A, 0, ftotal, &fgrid, c->search_margin));
            if (fa.xcd_affine) fgrid = std::max(8, (fgrid / 8) * 8);
            if (c->time_kernels) WOL_TRY(begin_timing(c, &ev, qs));
            if (!rows_kernel) hipLaunchKernelGGL(k_search_fast, dim3(fgrid), dim3(WOL_FAST_BLOCK), 0, qs, fa);
            else if (rows_block == WOL_ROWS_BLOCK_B) hipLaunchKernelGGL((k_search_rows<WOL_ROWS_BLOCK_B, WOL_ROWS_CAPP_B>), dim3(fgrid), dim3(WOL_ROWS_BLOCK_B), 0, qs, fa);
            else hipLaunchKernelGGL((k_search_rows<WOL_ROWS_BLOCK_A, WOL_ROWS_CAPP_A>), dim3(fgrid), dim3(WOL_ROWS_BLOCK_A), 0, qs, fa);
            if (c->time_kernels) WOL_TRY(end_timing(c, ev, qs));
            if (rows_kernel && c->check_props) {
                hipLaunchKernelGGL(k_check_props, dim3(1024), dim3(256), 0, qs, S.prop_key.as<int>(), S.prop.as<int4>(), (int64_t)F * n, n,
                                   c->prop_counters.as<unsigned long long>());
                c->prop_expected += (int64_t)bp.n_cell_frames * n;
            }
            if (pipe && qs != c->stream) {   // the certification stage follows on the caller-visible stream
                HIP_TRY(hipEventRecord(S.searched_ev, qs));
                HIP_TRY(hipStreamWaitEvent(c->stream, S.searched_ev, 0));
            }
            const int64_t wtotal = (int64_t)F * n;
            int ggrid = 0;
            const bool lean = fa.fast_q && !out.nn_idx && !out.nn_dist;
            const bool rec32 = bp.rec32;   // (float4 records: only behind the row kernel, see sort_batch)
            const bool idx = bp.idx;
            if (idx && lean) WOL_TRY(resident_grid(c, k_finish_fast<true, double4, true>, WOL_BLOCK, hist_lds, (wtotal + WOL_BLOCK - 1) / WOL_BLOCK, &ggrid));
            else if (idx) WOL_TRY(resident_grid(c, k_finish_fast<false, double4, true>, WOL_BLOCK, hist_lds, (wtotal + WOL_BLOCK - 1) / WOL_BLOCK, &ggrid));
            else if (rec32 && lean) WOL_TRY(resident_grid(c, k_finish_fast<true, float4>, WOL_BLOCK, hist_lds, (wtotal + WOL_BLOCK - 1) / WOL_BLOCK, &ggrid));
            else if (rec32) WOL_TRY(resident_grid(c, k_finish_fast<false, float4>, WOL_BLOCK, hist_lds, (wtotal + WOL_BLOCK - 1) / WOL_BLOCK, &ggrid));
            else if (lean) WOL_TRY(resident_grid(c, k_finish_fast<true>, WOL_BLOCK, hist_lds, (wtotal + WOL_BLOCK - 1) / WOL_BLOCK, &ggrid));
            else WOL_TRY(resident_grid(c, k_finish_fast<false>, WOL_BLOCK, hist_lds, (wtotal + WOL_BLOCK - 1) / WOL_BLOCK, &ggrid));
            if (fa.xcd_affine) ggrid = std::max(8, (ggrid / 8) * 8);
            fa.stat_part = nullptr; fa.stat_slots = 0;
            c->stat_slots = 0;
            if (lean && out.stat) {   // per-wave partial sums of the frame statistics (summed by k_stat_finalize)
                const int slots = (fa.xcd_affine ? ggrid / 8 : ggrid) * (WOL_BLOCK / 64);
                if ((size_t)slots * F * 32 <= ((size_t)256 << 20)) {
                    WOL_TRY(c->stat_part.ensure(sizeof(long long) * 4 * (size_t)slots * F));
                    HIP_TRY(hipMemsetAsync(c->stat_part.p, 0, sizeof(long long) * 4 * (size_t)slots * F, c->stream));
                    fa.stat_part = c->stat_part.as<long long>();
                    fa.stat_slots = slots;
                    c->stat_slots = slots;
                }
            }
            if (idx && lean) hipLaunchKernelGGL((k_finish_fast<true, double4, true>), dim3(ggrid), dim3(WOL_BLOCK), hist_lds, c->stream, fa);
            else if (idx) hipLaunchKernelGGL((k_finish_fast<false, double4, true>), dim3(ggrid), dim3(WOL_BLOCK), hist_lds, c->stream, fa);
            else if (rec32 && lean) hipLaunchKernelGGL((k_finish_fast<true, float4>), dim3(ggrid), dim3(WOL_BLOCK), hist_lds, c->stream, fa);
            else if (rec32) hipLaunchKernelGGL((k_finish_fast<false, float4>), dim3(ggrid), dim3(WOL_BLOCK), hist_lds, c->stream, fa);
            else if (lean) hipLaunchKernelGGL(k_finish_fast<true>, dim3(ggrid), dim3(WOL_BLOCK), hist_lds, c->stream, fa);
            else hipLaunchKernelGGL(k_finish_fast<false>, dim3(ggrid), dim3(WOL_BLOCK), hist_lds, c->stream, fa);
            int lgrid = 0;
            if (idx) {
                WOL_TRY(resident_grid(c, (k_search_exact_list<double4, true>), WOL_BLOCK, hist_lds, 1 << 20, &lgrid));
                hipLaunchKernelGGL((k_search_exact_list<double4, true>), dim3(lgrid), dim3(WOL_BLOCK), hist_lds, c->stream, a,
                                   c->slow_list.as<unsigned long long>(), fa.slow_count);
            } else if (rec32) {
                WOL_TRY(resident_grid(c, k_search_exact_list<float4>, WOL_BLOCK, hist_lds, 1 << 20, &lgrid));
                hipLaunchKernelGGL(k_search_exact_list<float4>, dim3(lgrid), dim3(WOL_BLOCK), hist_lds, c->stream, a,
                                   c->slow_list.as<unsigned long long>(), fa.slow_count);
            } else {
                WOL_TRY(resident_grid(c, k_search_exact_list<double4>, WOL_BLOCK, hist_lds, 1 << 20, &lgrid));
                hipLaunchKernelGGL(k_search_exact_list<double4>, dim3(lgrid), dim3(WOL_BLOCK), hist_lds, c->stream, a,
                                   c->slow_list.as<unsigned long long>(), fa.slow_count);
            }
        } else {
            if (c->time_kernels) WOL_TRY(begin_timing(c, &ev));
            hipLaunchKernelGGL(k_search_exact, dim3(grid), dim3(WOL_BLOCK), hist_lds, c->stream, a);
            if (c->time_kernels) WOL_TRY(end_timing(c, ev));
        }
        HIP_TRY(hipEventRecord(S.free_ev, c->stream));
        S.used = true;
    }
    for (int f : brute_frames) {
        PointsView<T> pv{fv.base + (int64_t)f * fv.frame_stride, fv.atom_stride, n};
        PointsView<T> sv = sub ? PointsView<T>{sub, sub_stride, ns} : pv;
        const Box3 b = make_box(boxes_host + 3 * (size_t)f);
        std::pair<hipEvent_t, hipEvent_t> ev;
        if (c->time_kernels) WOL_TRY(begin_timing(c, &ev));
        hipLaunchKernelGGL(k_brute_q<T>, dim3((ns + WOL_BLOCK - 1) / WOL_BLOCK), dim3(WOL_BLOCK), hist_lds,
                           c->stream, pv, sv, b, lowsq, highsq, out, (int64_t)f * ns, f);
        if (c->time_kernels) WOL_TRY(end_timing(c, ev));
    }
    HIP_TRY(hipGetLastError());
    return WOL_OK;
}

// exclusive scan of the cell counters of F frames (tables of up to ncell_max cells)
int launch_cell_scan(wol_ctx* c, hipStream_t st, const FrameGrid* grids, int* cells, int F, int ncell_max) {
    if (ncell_max <= 32768 || ncell_max > 4096 * 1024) {
        hipLaunchKernelGGL(k_cell_scan, dim3(F), dim3(1024), 0, st, grids, cells);
        return WOL_OK;
    }
    const int ntile = (ncell_max + 4095) / 4096;
    WOL_TRY(c->scan_tot.ensure(sizeof(int) * (size_t)F * ntile));
    int* tt = c->scan_tot.as<int>();
    hipLaunchKernelGGL(k_cell_scan_tiles, dim3(ntile, F), dim3(1024), 0, st, grids, cells, tt, ntile);
    hipLaunchKernelGGL(k_cell_scan_totals, dim3(F), dim3(1024), 0, st, grids, cells, tt, ntile);
    hipLaunchKernelGGL(k_cell_scan_add, dim3(ntile, F), dim3(1024), 0, st, grids, cells, tt, ntile);
    return WOL_OK;
}

// Single-frame set-up shared by wol_cos_angs / wol_lsi_frame: upload, bin, cell-sort; the grid is
// built for the given reach so walking cells [-lo, hi]^3 sees every neighbour inside it.
// Where a caller's array of reals lives and how it is stored: the per-frame calls accept double or float storage in host
// or device memory (include/wol.h: WOL_F64 / WOL_F32, WOL_MEM_HOST / WOL_MEM_DEVICE); the arithmetic is always the
// reference's fp64 (floats are promoted exactly on the way in, results rounded once on the way out).
struct RealSrc { const void* p; int dtype; int mem; };

__global__ void k_cvt_f32_f64(const float* __restrict__ src, double* __restrict__ dst, int64_t count) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) dst[i] = (double)src[i];
}
__global__ void k_cvt_f64_f32(const double* __restrict__ src, float* __restrict__ dst, int64_t count) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) dst[i] = (float)src[i];
}
__global__ void k_int_to_f64(const int* __restrict__ src, double* __restrict__ dst, int64_t count) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) dst[i] = (double)src[i];
}
__global__ void k_add_u64(const unsigned long long* __restrict__ src, unsigned long long* __restrict__ dst, int count) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) dst[i] += src[i];
}

// `count` reals of the caller -> doubles at dev_dst (device), on the context's stream
int fetch_f64(wol_ctx* c, RealSrc src, size_t count, double* dev_dst) {
    if (count == 0) return WOL_OK;
    const unsigned blocks = (unsigned)((count + 255) / 256);
    if (src.dtype == WOL_F64) {
        HIP_TRY(hipMemcpyAsync(dev_dst, src.p, sizeof(double) * count,
                               src.mem == WOL_MEM_HOST ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice, c->stream));
        return WOL_OK;
    }
    const float* f = static_cast<const float*>(src.p);
    if (src.mem == WOL_MEM_HOST) {
        WOL_TRY(c->cvt_in.ensure(sizeof(float) * count));
        HIP_TRY(hipMemcpyAsync(c->cvt_in.p, src.p, sizeof(float) * count, hipMemcpyHostToDevice, c->stream));
        f = c->cvt_in.as<float>();
    }
    hipLaunchKernelGGL(k_cvt_f32_f64, dim3(blocks), dim3(256), 0, c->stream, f, dev_dst, (int64_t)count);
    return WOL_OK;
}

// `count` doubles at dev_src (device) -> the caller's array, on the context's stream (host destinations are complete
// once the stream has been synchronised); `slot` picks one of the conversion staging buffers so several float
// outputs of one call can be in flight
int deliver_f64(wol_ctx* c, const double* dev_src, void* dst, size_t count, int dtype, int mem, int slot = 0) {
    if (count == 0 || !dst) return WOL_OK;
    if (dtype == WOL_F64) {
        HIP_TRY(hipMemcpyAsync(dst, dev_src, sizeof(double) * count,
                               mem == WOL_MEM_HOST ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice, c->stream));
        return WOL_OK;
    }
    const unsigned blocks = (unsigned)((count + 255) / 256);
    if (mem == WOL_MEM_DEVICE) {
        hipLaunchKernelGGL(k_cvt_f64_f32, dim3(blocks), dim3(256), 0, c->stream, dev_src, static_cast<float*>(dst), (int64_t)count);
        return WOL_OK;
    }
    DevBuf& b = c->cvt_out[slot & 3];
    WOL_TRY(b.ensure(sizeof(float) * count));
    hipLaunchKernelGGL(k_cvt_f64_f32, dim3(blocks), dim3(256), 0, c->stream, dev_src, b.as<float>(), (int64_t)count);
    HIP_TRY(hipMemcpyAsync(dst, b.p, sizeof(float) * count, hipMemcpyDeviceToHost, c->stream));
    return WOL_OK;
}

int check_real(int dtype, int mem) {
    if (dtype != WOL_F64 && dtype != WOL_F32) return fail(WOL_ERR_INVALID, "bad dtype (WOL_F64 or WOL_F32)");
    if (mem != WOL_MEM_HOST && mem != WOL_MEM_DEVICE) return fail(WOL_ERR_INVALID, "bad mem space");
    return WOL_OK;
}

int shell_setup(wol_ctx* c, RealSrc pos, int n, RealSrc sub, int ns, const double* box,
                double reach, ShellArgs* sa) {
    if (c->grid_copy_pending) {
        HIP_TRY(hipEventSynchronize(c->grid_ev));
        c->grid_copy_pending = false;
    }
    wol_ctx::SortSet& S = c->sets[0];
    if (c->sort_stream) HIP_TRY(hipStreamSynchronize(c->sort_stream));  // rare path: keep it simple
    if (c->search_stream) HIP_TRY(hipStreamSynchronize(c->search_stream));
    if (c->grid_stream) HIP_TRY(hipStreamSynchronize(c->grid_stream));
    c->shell_dirty = true;
    WOL_TRY(c->grids_host.ensure(sizeof(FrameGrid)));
    FrameGrid* gh = c->grids_host.as<FrameGrid>();
    // cells as wide as the reach (27-cell walk) for short reaches, half the reach (125 cells) for long ones: the
    // smaller candidate volume of the two that keeps cells above 2.5 A (empty cells still cost a table entry)
    const bool ok = make_grid(box, n, reach < 5.0 ? reach * 1.001 : std::fmax(reach * 0.5, 2.5), reach, gh);
    if (!ok) { gh->ncell = 1; for (int k = 0; k < 3; ++k) gh->nc[k] = 1; }
    gh->cell_base = 0;
    const size_t pe = 3 * (size_t)n, se = sub.p ? 3 * (size_t)ns : 0;
    WOL_TRY(c->in_stage.ensure(sizeof(double) * (pe + se)));
    WOL_TRY(S.grids.ensure(sizeof(FrameGrid)));
    WOL_TRY(S.cells.ensure(sizeof(int) * (size_t)(gh->ncell + 1)));
    WOL_TRY(S.sorted.ensure(sizeof(double4) * (size_t)n));
    WOL_TRY(c->cellid.ensure(sizeof(int) * (size_t)n));
    WOL_TRY(c->rank.ensure(sizeof(int) * (size_t)n));
    WOL_TRY(fetch_f64(c, pos, pe, c->in_stage.as<double>()));
    if (sub.p) WOL_TRY(fetch_f64(c, sub, se, c->in_stage.as<double>() + pe));
    HIP_TRY(hipMemcpyAsync(S.grids.p, gh, sizeof(FrameGrid), hipMemcpyHostToDevice, c->stream));
    if (!c->grid_ev) HIP_TRY(hipEventCreateWithFlags(&c->grid_ev, hipEventDisableTiming));
    HIP_TRY(hipEventRecord(c->grid_ev, c->stream));
    c->grid_copy_pending = true;
    HIP_TRY(hipMemsetAsync(S.cells.p, 0, sizeof(int) * (size_t)(gh->ncell + 1), c->stream));
    FramesView<double> fv{c->in_stage.as<double>(), 3, 0, n};
    const dim3 gb((n + WOL_BLOCK - 1) / WOL_BLOCK, 1);
    hipLaunchKernelGGL(k_bin_count<double>, gb, dim3(WOL_BLOCK), 0, c->stream, fv, S.grids.as<FrameGrid>(),
                       S.cells.as<int>(), c->cellid.as<int>(), c->rank.as<int>());
    WOL_TRY(launch_cell_scan(c, c->stream, S.grids.as<FrameGrid>(), S.cells.as<int>(), 1, gh->ncell));
    hipLaunchKernelGGL(k_scatter<double>, gb, dim3(WOL_BLOCK), 0, c->stream, fv, S.grids.as<FrameGrid>(),
                       S.cells.as<int>(), c->cellid.as<int>(), c->rank.as<int>(), S.sorted.as<double4>(),
                       (float4*)nullptr);
    // cells in original-index order: the shell kernels' floating-point sums become reproducible
    hipLaunchKernelGGL(k_cell_order, dim3((gh->ncell + WOL_BLOCK - 1) / WOL_BLOCK), dim3(WOL_BLOCK), 0, c->stream,
                       S.grids.as<FrameGrid>(), S.cells.as<int>(), S.sorted.as<double4>());
    sa->grid = S.grids.as<FrameGrid>();
    sa->sorted = S.sorted.as<double4>();
    sa->cell_start = S.cells.as<int>();
    sa->n = n;
    sa->sub = sub.p ? c->in_stage.as<double>() + pe : nullptr;
    sa->ns = sub.p ? ns : n;
    sa->use_cells = ok ? 1 : 0;
    return WOL_OK;
}

int shell_setup(wol_ctx* c, const double* pos, int n, const double* sub, int ns, const double* box,
                double reach, ShellArgs* sa) {
    return shell_setup(c, RealSrc{pos, WOL_F64, WOL_MEM_HOST}, n, RealSrc{sub, WOL_F64, WOL_MEM_HOST}, ns, box, reach, sa);
}

int check_ctx(wol_ctx* c) {
    if (!c) return fail(WOL_ERR_INVALID, "null context");
    HIP_TRY(hipSetDevice(c->device));
    return WOL_OK;
}

// ---------------------------------------------------------------------------------------------
// Frames of a trajectory call, batch by batch, as device-resident views.  Device frames: views into the caller's array.
// Host frames: double-buffered staging; the copy of batch b + 1 crosses the link while batch b's kernels run.
// What crosses the link (option "host_pack"):
//   0  the frames as they lie in host memory, i.e. whole interleaved [O, H, H, EP] frames (one linear copy per batch);
//   1  the O rows only: host threads (option "host_threads") pack the 24-byte rows the path reads -- what the reference's
//      caller does with pos[watInds] (orderParam_lib.py:1462) -- into pinned staging, a quarter of a TIP4P frame's bytes;
//   2  (default) the same, stored as float32 when EVERY coordinate of the batch is exactly representable as one (frames
//      read from single-precision trajectory files are): promoted back exactly on the device, so results are bit for bit
//      those of the fp64 frames at half the bytes again.  The first inexact value ends the attempt for the rest of the call.
// Frames that are O-only already (atom stride 3) skip the packing unless narrowing applies.
// ---------------------------------------------------------------------------------------------
struct FrameBatchView {
    const void* base = nullptr;   // device
    bool f32 = false;             // element type of `base` (float or double)
    int64_t atom_stride = 3, frame_stride = 0;
};

template <typename T>
struct FrameFeeder {
    wol_ctx* c = nullptr;
    const T* frames = nullptr;
    int64_t atom_stride = 3, frame_stride = 0;
    int T_frames = 0, n = 0, Fb = 0, nbatch = 0;
    bool host = false, pack = false, narrow = false;
    std::vector<int> bstart;
    bool batch_f32[2] = {false, false};

    size_t batch_elems(int F) const { return (size_t)(F - 1) * frame_stride + (size_t)(n - 1) * atom_stride + 3; }

    // batches of the call: Fb frames each (0 = automatic, bytes_per_water = intermediate device bytes a water costs)
    int plan(wol_ctx* ctx, const T* fr, int64_t as, int64_t fs, int Tn, int nw, bool is_host, double bytes_per_water) {
        c = ctx; frames = fr; atom_stride = as; frame_stride = fs; T_frames = Tn; n = nw; host = is_host;
        pack = host && c->host_pack >= 1 && (atom_stride > 3 || (c->host_pack >= 2 && std::is_same<T, double>::value));
        narrow = pack && c->host_pack >= 2 && std::is_same<T, double>::value;
        if (narrow) {
            // a look at the first waters of the first frame decides whether narrowing is worth trying at all (every batch is
            // still checked in full while it is packed)
            const int probe = std::min(n, 4096);
            for (int w = 0; w < probe && narrow; ++w) {
                const T* r = frames + (int64_t)w * atom_stride;
                narrow = ((T)(float)r[0] == r[0]) && ((T)(float)r[1] == r[1]) && ((T)(float)r[2] == r[2]);
            }
            if (!narrow && atom_stride == 3) pack = false;    // O-only frames that are not single precision: copied as they lie
        }
        Fb = c->frames_per_batch;
        if (Fb <= 0) {
            if (host) {
                // host frames: batches of ~256 MB of link traffic (at least 64 frames: the streaming sort wants that many), so
                // the first copy -- the only one no kernel runs under -- stays a small part of the job
                // (packed frames: shorter batches, ~64 MB -- the first batch's packing and copy run under nothing)
                const double frame_bytes = pack ? 3.0 * n * sizeof(T) : (double)frame_stride * sizeof(T);
                Fb = pack ? (int)std::fmin(4096.0, std::fmax(32.0, 64.0e6 / frame_bytes)) : (int)std::fmin(4096.0, std::fmax(64.0, 256.0e6 / frame_bytes));
                const double per_frame = (double)n * bytes_per_water + 2.0 * frame_bytes;
                Fb = (int)std::fmin((double)Fb, std::fmax(1.0, 6.0e9 / per_frame));
            } else {
                // device-resident frames: up to 125 M waters per batch (about 30 GB of intermediate arrays in three sets, of
                // 288 GB) -- a 32-frame block of a million-water box is one batch, not 24 + 8
                const double per_frame = (double)n * bytes_per_water;
                Fb = (int)std::fmin((double)T_frames, std::fmax(1.0, 8.0e9 / per_frame));
            }
            if (Fb >= 8 && Fb < T_frames) Fb = (Fb / 8) * 8;
        }
        Fb = std::max(1, std::min(Fb, T_frames));
        // batch b = frames [bstart[b], bstart[b + 1]).  Host frames: what the call costs beyond the copies is the kernels of
        // the LAST batch (nothing crosses the link under them), so the tail of a several-batch call is cut off as a batch of
        // its own of at most 32 frames
        bstart.clear();
        for (int f0 = 0; f0 < T_frames; f0 += Fb) bstart.push_back(f0);
        bstart.push_back(T_frames);
        if (host && c->frames_per_batch <= 0 && bstart.size() >= 3) {
            const int last = T_frames - bstart[bstart.size() - 2];
            if (last > 48) bstart.insert(bstart.end() - 1, T_frames - 32);
        }
        nbatch = (int)bstart.size() - 1;
        return WOL_OK;
    }

    // the O rows of frames [f0, f0 + F) -> dst (packed [F][n][3]); narrowed to float when asked and exact.  Returns false
    // when narrowing was asked for and a value is not a float (dst then holds garbage).
    bool pack_rows(int f0, int F, void* dst, bool to_f32) const {
        const int64_t rows = (int64_t)F * n;
        int nt = c->host_threads > 0 ? c->host_threads : (int)std::min(16u, std::max(1u, std::thread::hardware_concurrency()));
        nt = (int)std::max<int64_t>(1, std::min<int64_t>(nt, rows / 4096));
        std::atomic<bool> exact(true);
        auto work = [&](int64_t r0, int64_t r1) {
            int f = (int)(r0 / n);
            int64_t w = r0 - (int64_t)f * n;
            const T* src = frames + (int64_t)(f0 + f) * frame_stride + w * atom_stride;
            bool ok = true;
            if (to_f32) {
                float* d = static_cast<float*>(dst) + 3 * r0;
                for (int64_t r = r0; r < r1; ++r) {
                    const float x = (float)src[0], y = (float)src[1], z = (float)src[2];
                    ok = ok && ((T)x == src[0]) && ((T)y == src[1]) && ((T)z == src[2]);
                    d[0] = x; d[1] = y; d[2] = z;
                    d += 3; src += atom_stride;
                    if (++w == n) { w = 0; ++f; src = frames + (int64_t)(f0 + f) * frame_stride; if (!ok) break; }
                }
            } else {
                T* d = static_cast<T*>(dst) + 3 * r0;
                for (int64_t r = r0; r < r1; ++r) {
                    d[0] = src[0]; d[1] = src[1]; d[2] = src[2];
                    d += 3; src += atom_stride;
                    if (++w == n) { w = 0; ++f; src = frames + (int64_t)(f0 + f) * frame_stride; }
                }
            }
            if (!ok) exact.store(false, std::memory_order_relaxed);
        };
        if (nt <= 1) {
            work(0, rows);
        } else {
            std::vector<std::thread> th;
            th.reserve((size_t)nt - 1);
            for (int i = 1; i < nt; ++i) th.emplace_back(work, rows * i / nt, rows * (i + 1) / nt);
            work(0, rows / nt);
            for (auto& t : th) t.join();
        }
        return exact.load();
    }

    int enqueue_copy(int b) {
        const int f0 = bstart[b], F = bstart[b + 1] - f0, k = b & 1;
        if (!pack) {
            if (b >= 2) HIP_TRY(hipStreamWaitEvent(c->copy_stream, c->consumed_ev[k], 0));   // batch b-2 is done with this set
            HIP_TRY(hipMemcpyAsync(c->host_stage[k].p, frames + (int64_t)f0 * frame_stride, batch_elems(F) * sizeof(T),
                                   hipMemcpyHostToDevice, c->copy_stream));
            HIP_TRY(hipEventRecord(c->copied_ev[k], c->copy_stream));
            c->last_h2d_bytes += (double)(batch_elems(F) * sizeof(T));
            batch_f32[k] = !std::is_same<T, double>::value;
            return WOL_OK;
        }
        // the pinned set is free once the copy that last read it (batch b - 2) has left it
        if (b >= 2) HIP_TRY(hipEventSynchronize(c->copied_ev[k]));
        bool as_f32 = !std::is_same<T, double>::value, packed = false;
        if (narrow) {
            if (pack_rows(f0, F, c->pack_stage[k].p, true)) { as_f32 = true; packed = true; }
            else narrow = false;     // this trajectory is not single precision: stop trying
        }
        if (!packed) (void)pack_rows(f0, F, c->pack_stage[k].p, false);
        const size_t bytes = (size_t)F * n * 3 * (as_f32 ? sizeof(float) : sizeof(T));
        if (b >= 2) HIP_TRY(hipStreamWaitEvent(c->copy_stream, c->consumed_ev[k], 0));
        HIP_TRY(hipMemcpyAsync(c->host_stage[k].p, c->pack_stage[k].p, bytes, hipMemcpyHostToDevice, c->copy_stream));
        HIP_TRY(hipEventRecord(c->copied_ev[k], c->copy_stream));
        c->last_h2d_bytes += (double)bytes;
        batch_f32[k] = as_f32;
        return WOL_OK;
    }

    int start() {
        if (!host) return WOL_OK;
        if (!c->copy_stream) HIP_TRY(hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
        const int Fmax = std::min(Fb, T_frames);
        for (int k = 0; k < 2; ++k) {
            if (!c->copied_ev[k]) HIP_TRY(hipEventCreateWithFlags(&c->copied_ev[k], hipEventDisableTiming));
            if (!c->consumed_ev[k]) HIP_TRY(hipEventCreateWithFlags(&c->consumed_ev[k], hipEventDisableTiming));
            if (k < nbatch) {
                WOL_TRY(c->host_stage[k].ensure(pack ? (size_t)Fmax * n * 3 * sizeof(T) : batch_elems(Fmax) * sizeof(T)));
                if (pack) WOL_TRY(c->pack_stage[k].ensure((size_t)Fmax * n * 3 * sizeof(T)));
            }
        }
        // earlier work on the context's stream may still read the staging sets (a previous call's last batches)
        HIP_TRY(hipStreamSynchronize(c->stream));
        HIP_TRY(hipStreamSynchronize(c->copy_stream));
        c->last_h2d_bytes = 0.0;
        return enqueue_copy(0);
    }

    // batch b as a device view; its kernels may be enqueued on c->stream right away
    int acquire(int b, FrameBatchView* v) {
        const int f0 = bstart[b];
        if (host) {
            HIP_TRY(hipStreamWaitEvent(c->stream, c->copied_ev[b & 1], 0));
            v->base = c->host_stage[b & 1].p;
            v->f32 = batch_f32[b & 1];
            if (pack) { v->atom_stride = 3; v->frame_stride = 3 * (int64_t)n; }
            else { v->atom_stride = atom_stride; v->frame_stride = frame_stride; }
        } else {
            v->base = frames + (int64_t)f0 * frame_stride;
            v->f32 = !std::is_same<T, double>::value;
            v->atom_stride = atom_stride; v->frame_stride = frame_stride;
        }
        return WOL_OK;
    }

    // batch b's kernels are enqueued: its staging set may be refilled behind them, the next batch may start to travel
    int release(int b) {
        if (!host) return WOL_OK;
        HIP_TRY(hipEventRecord(c->consumed_ev[b & 1], c->stream));
        // (a copy from pageable memory, or the packing, blocks the host: this batch's kernels are enqueued already)
        if (b + 1 < nbatch) WOL_TRY(enqueue_copy(b + 1));
        return WOL_OK;
    }
};

// populations of a trajectory call (tetOrderCalc / threeBodyCalc / lsiCalc with subInds): query points by position
struct PopQuery {
    const void* pos = nullptr;       // [total][3] reals of the call's dtype, in the call's memory space
    const int64_t* off = nullptr;    // host [T * npops + 1]
    int npops = 0;
};

// stages the population points (as doubles) and their offsets on the device; *total = number of points
int stage_populations(wol_ctx* c, const PopQuery& pq, int T_frames, int dtype, int mem, const double** d_pts, const long long** d_off, int64_t* total) {
    *d_pts = nullptr; *d_off = nullptr; *total = 0;
    if (pq.npops <= 0) return WOL_OK;
    if (!pq.off) return fail(WOL_ERR_INVALID, "population offsets must not be NULL");
    const size_t noff = (size_t)T_frames * pq.npops + 1;
    if (pq.off[0] != 0) return fail(WOL_ERR_INVALID, "population offsets must start at 0");
    for (size_t i = 1; i < noff; ++i)
        if (pq.off[i] < pq.off[i - 1]) return fail(WOL_ERR_INVALID, "population offsets must not decrease");
    *total = pq.off[noff - 1];
    if (*total > 0 && !pq.pos) return fail(WOL_ERR_INVALID, "population points must not be NULL");
    WOL_TRY(c->pop_off.ensure(sizeof(long long) * noff));
    HIP_TRY(hipMemcpyAsync(c->pop_off.p, pq.off, sizeof(long long) * noff, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));    // (the caller's offsets array may be a temporary)
    *d_off = c->pop_off.as<long long>();
    if (*total > 0) {
        if (dtype == WOL_F64 && mem == WOL_MEM_DEVICE) {
            *d_pts = static_cast<const double*>(pq.pos);
        } else {
            WOL_TRY(c->pop_pts.ensure(sizeof(double) * 3 * (size_t)*total));
            WOL_TRY(fetch_f64(c, RealSrc{pq.pos, dtype, mem}, 3 * (size_t)*total, c->pop_pts.as<double>()));
            *d_pts = c->pop_pts.as<double>();
        }
    }
    return WOL_OK;
}

template <typename U>
FramesView<U> view_as(const FrameBatchView& v, int n) {
    FramesView<U> fv;
    fv.base = static_cast<const U*>(v.base);
    fv.atom_stride = v.atom_stride;
    fv.frame_stride = v.frame_stride;
    fv.n = n;
    return fv;
}

template <typename T>
int q_traj_impl(wol_ctx* c, const T* frames, int64_t atom_stride, int64_t frame_stride, int T_frames,
                int n, const double* boxes, double low, double high, double* mean, double* var,
                int64_t* qhist, int qbins, int64_t* dhist, int dbins, double dbinwidth, T* q_out, int mem,
                const PopQuery* pq = nullptr, double* pop_mean = nullptr, double* pop_var = nullptr, int64_t* pop_qhist = nullptr,
                T* pop_q = nullptr) {
    WOL_TRY(check_ctx(c));
    if (!frames || !boxes) return fail(WOL_ERR_INVALID, "frames/boxes must not be NULL");
    if (T_frames < 0 || n <= 0) return fail(WOL_ERR_INVALID, "need T >= 0 and n > 0 (got T=%d n=%d)", T_frames, n);
    if (atom_stride < 3) return fail(WOL_ERR_INVALID, "atom_stride must be >= 3");
    if (frame_stride < (int64_t)(n - 1) * atom_stride + 3 && T_frames > 1)
        return fail(WOL_ERR_INVALID, "frame_stride smaller than one frame");
    if (qhist && qbins <= 0) return fail(WOL_ERR_INVALID, "qbins must be > 0");
    if (dhist && (dbins <= 0 || !(dbinwidth > 0.0))) return fail(WOL_ERR_INVALID, "dbins/dbinwidth must be > 0");
    if ((qhist ? qbins : 0) + (dhist ? dbins : 0) > 12000) return fail(WOL_ERR_INVALID, "too many histogram bins for LDS");
    if (mem != WOL_MEM_HOST && mem != WOL_MEM_DEVICE) return fail(WOL_ERR_INVALID, "bad mem space");
    const int npops = pq ? pq->npops : 0;
    if (npops < 0) return fail(WOL_ERR_INVALID, "npops must be >= 0");
    if (npops > 0 && pop_qhist && qbins <= 0) return fail(WOL_ERR_INVALID, "qbins must be > 0");
    if (T_frames == 0) return WOL_OK;
    const bool host = (mem == WOL_MEM_HOST);
    const int dtype = std::is_same<T, double>::value ? WOL_F64 : WOL_F32;

    FrameFeeder<T> feed;
    WOL_TRY(feed.plan(c, frames, atom_stride, frame_stride, T_frames, n, host, 64.0));
    const int Fb = feed.Fb, nbatch = feed.nbatch;

    // populations: points and offsets on the device, integer accumulators per (frame, population) segment
    const double* d_pts = nullptr;
    const long long* d_poff = nullptr;
    int64_t npts = 0;
    const int npops1 = npops + 1;
    if (npops > 0) {
        WOL_TRY(stage_populations(c, *pq, T_frames, dtype, mem, &d_pts, &d_poff, &npts));
        const size_t sb = sizeof(unsigned long long) * 8 * (size_t)T_frames * npops1, hb = sizeof(unsigned long long) * (size_t)npops * std::max(qbins, 1);
        WOL_TRY(c->pop_stat.ensure(sb));
        WOL_TRY(c->pop_hist.ensure(hb));
        HIP_TRY(hipMemsetAsync(c->pop_stat.p, 0, sb, c->stream));
        HIP_TRY(hipMemsetAsync(c->pop_hist.p, 0, hb, c->stream));
        if (pop_q && npts > 0) WOL_TRY(c->pop_out.ensure(sizeof(double) * (size_t)npts));
    }

    // device histograms (uint64 == the caller's int64 bit pattern for non-negative counts)
    unsigned long long* d_qh = nullptr;
    unsigned long long* d_dh = nullptr;
    if (host && (qhist || dhist)) {
        WOL_TRY(c->hist_stage.ensure(sizeof(unsigned long long) * (size_t)(qbins + dbins + 2)));
        HIP_TRY(hipMemsetAsync(c->hist_stage.p, 0, sizeof(unsigned long long) * (size_t)(qbins + dbins + 2), c->stream));
        if (qhist) d_qh = c->hist_stage.as<unsigned long long>();
        if (dhist) d_dh = c->hist_stage.as<unsigned long long>() + qbins;
    } else {
        d_qh = reinterpret_cast<unsigned long long*>(qhist);
        d_dh = reinterpret_cast<unsigned long long*>(dhist);
    }
    const bool need_stats = mean || var;
    const bool q_direct = !host && q_out && std::is_same<T, double>::value;  // q written straight to the caller
    double* d_stats = nullptr;
    if (host && need_stats) {
        WOL_TRY(c->small.ensure(sizeof(double) * 2 * (size_t)T_frames));
        d_stats = c->small.as<double>();
    }
    WOL_TRY(feed.start());

    if (c->overlap && !host && !c->inputs_resident) {
        // overlapped batches sort on a side stream: mark, once per call and whether or not the first batch is piped, the
        // point of the caller's stream behind which the frames are complete
        if (!c->in_ev) HIP_TRY(hipEventCreateWithFlags(&c->in_ev, hipEventDisableTiming));
        HIP_TRY(hipEventRecord(c->in_ev, c->stream));
    }
    for (int b = 0; b < nbatch; ++b) {
        const int f0 = feed.bstart[b];
        const int F = feed.bstart[b + 1] - f0;
        FrameBatchView bv;
        WOL_TRY(feed.acquire(b, &bv));
        QOut out;
        memset(&out, 0, sizeof out);
        double* qd = nullptr;
        if (q_direct) {
            qd = reinterpret_cast<double*>(q_out) + (int64_t)f0 * n;
        } else if (q_out && std::is_same<T, double>::value) {
            WOL_TRY(c->qbuf.ensure(sizeof(double) * (size_t)std::min(Fb, T_frames) * n));
            qd = c->qbuf.as<double>();
        }
        out.q = qd;  // per-water q is only materialised when the caller asks for it
        float* q32d = nullptr;
        if (q_out && !std::is_same<T, double>::value) {
            if (host) {
                WOL_TRY(c->out_stage.ensure(sizeof(float) * (size_t)std::min(Fb, T_frames) * n));
                q32d = c->out_stage.as<float>();
            } else {
                q32d = reinterpret_cast<float*>(q_out) + (int64_t)f0 * n;
            }
        }
        out.q32 = q32d;
        out.qhist = d_qh; out.qbins = qbins;
        out.dhist = d_dh; out.dbins = dbins; out.dbinwidth = dbinwidth;
        if (need_stats) {
            WOL_TRY(c->stat.ensure(sizeof(unsigned long long) * 4 * (size_t)std::min(Fb, T_frames)));
            HIP_TRY(hipMemsetAsync(c->stat.p, 0, sizeof(unsigned long long) * 4 * (size_t)F, c->stream));
            out.stat = c->stat.as<unsigned long long>();
        }
        const int pipe_mode = host ? 0 : (b == 0 ? 1 : 2);
        // (a packed batch narrowed to float32 runs the float instantiation: exact promotion, identical results)
        if (bv.f32) WOL_TRY(run_q_batch<float>(c, view_as<float>(bv, n), F, boxes + 3 * (size_t)f0, nullptr, 0, n, low, high, out, pipe_mode, npops > 0,
                                               npops == 0));   // (the population kernels read double4 records)
        else WOL_TRY(run_q_batch<double>(c, view_as<double>(bv, n), F, boxes + 3 * (size_t)f0, nullptr, 0, n, low, high, out, pipe_mode, npops > 0,
                                         npops == 0));
        if (need_stats) {
            double* dm = host ? d_stats + f0 : (mean ? mean + f0 : nullptr);
            double* dv = host ? d_stats + T_frames + f0 : (var ? var + f0 : nullptr);
            hipLaunchKernelGGL(k_stat_finalize, dim3(F), dim3(64), 0, c->stream,
                               c->stat.as<unsigned long long>(), c->stat_slots ? c->stat_part.as<long long>() : nullptr,
                               c->stat_slots, F, n, dm, dv);
        }
        if (npops > 0 && feed.bstart[b + 1] > f0 && pq->off[(size_t)(f0 + F) * npops] > pq->off[(size_t)f0 * npops] && c->last_set) {
            // the populations' points of these frames against the batch just sorted (its arrays stay valid until the
            // set is reused, behind this stream)
            wol_ctx::SortSet& S = *c->last_set;
            TrajShellArgs ta;
            memset(&ta, 0, sizeof ta);
            ta.grids = S.grids.as<FrameGrid>(); ta.sorted = S.sorted.as<double4>(); ta.cell_start = S.cells.as<int>();
            ta.n = n; ta.F = F; ta.f0 = f0; ta.npops = npops;
            ta.lowsq = low * low; ta.highsq = high * high;
            ta.pts = d_pts; ta.pt_off = d_poff;
            ta.hist = c->pop_hist.as<unsigned long long>(); ta.stat = c->pop_stat.as<unsigned long long>();
            const int64_t pts_here = pq->off[(size_t)(f0 + F) * npops] - pq->off[(size_t)f0 * npops];
            const int grid = (int)std::max<int64_t>(1, std::min<int64_t>((pts_here + 3) / 4, 8 * (int64_t)c->num_cus));
            hipLaunchKernelGGL(k_q_traj_points, dim3(grid), dim3(WOL_BLOCK), 0, c->stream, ta, high, qbins,
                               (pop_q && npts > 0) ? c->pop_out.as<double>() : nullptr);
        }
        WOL_TRY(feed.release(b));
        if (host && q_out) {   // behind this batch's kernels on the stream; the staging buffer is reused in stream order
            if (std::is_same<T, double>::value)
                HIP_TRY(hipMemcpyAsync(q_out + (int64_t)f0 * n, qd, sizeof(double) * (size_t)F * n, hipMemcpyDeviceToHost, c->stream));
            else
                HIP_TRY(hipMemcpyAsync(q_out + (int64_t)f0 * n, q32d, sizeof(float) * (size_t)F * n, hipMemcpyDeviceToHost, c->stream));
        }
    }
    if (npops > 0) {
        // per (frame, population): np.mean / np.var of the population's q (NaN for an empty one, as numpy); histograms
        // ACCUMULATE into the caller's [npops][qbins] array like the all-waters histogram
        const int nseg = T_frames * npops1;
        WOL_TRY(c->pop_res.ensure(sizeof(double) * 2 * (size_t)T_frames * npops));
        double* rm = c->pop_res.as<double>();
        double* rv = rm + (size_t)T_frames * npops;
        hipLaunchKernelGGL(k_traj_finalize, dim3((nseg + 255) / 256), dim3(256), 0, c->stream, c->pop_stat.as<unsigned long long>(), nseg, 0,
                           npops1, 1, 1, (long long*)nullptr, rm, rv);
        const size_t nb = sizeof(double) * (size_t)T_frames * npops;
        const hipMemcpyKind kind = host ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice;
        if (pop_mean) HIP_TRY(hipMemcpyAsync(pop_mean, rm, nb, kind, c->stream));
        if (pop_var) HIP_TRY(hipMemcpyAsync(pop_var, rv, nb, kind, c->stream));
        if (pop_q && npts > 0) WOL_TRY(deliver_f64(c, c->pop_out.as<double>(), pop_q, (size_t)npts, dtype, mem, 3));
        if (pop_qhist && !host)
            hipLaunchKernelGGL(k_add_u64, dim3((npops * qbins + 255) / 256), dim3(256), 0, c->stream, c->pop_hist.as<unsigned long long>(),
                               reinterpret_cast<unsigned long long*>(pop_qhist), npops * qbins);
    }
    if (host) {
        if (mean) HIP_TRY(hipMemcpyAsync(mean, d_stats, sizeof(double) * (size_t)T_frames, hipMemcpyDeviceToHost, c->stream));
        if (var) HIP_TRY(hipMemcpyAsync(var, d_stats + T_frames, sizeof(double) * (size_t)T_frames, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        if (qhist || dhist) {
            std::vector<unsigned long long> h((size_t)qbins + dbins + 2);
            HIP_TRY(hipMemcpy(h.data(), c->hist_stage.p, sizeof(unsigned long long) * h.size(), hipMemcpyDeviceToHost));
            if (qhist) for (int k = 0; k < qbins; ++k) qhist[k] += (int64_t)h[k];
            if (dhist) for (int k = 0; k < dbins; ++k) dhist[k] += (int64_t)h[(size_t)qbins + k];
        }
        if (npops > 0 && pop_qhist) {
            std::vector<unsigned long long> h((size_t)npops * qbins);
            HIP_TRY(hipMemcpy(h.data(), c->pop_hist.p, sizeof(unsigned long long) * h.size(), hipMemcpyDeviceToHost));
            for (size_t k = 0; k < h.size(); ++k) pop_qhist[k] += (int64_t)h[k];
        }
    }
    HIP_TRY(hipGetLastError());
    return WOL_OK;
}

// ---------------------------------------------------------------------------------------------
// threeBodyCalc / lsiCalc frame loops (wol_cos_angs_traj, wol_lsi_traj): batches through the feeder and the sort stage,
// the kernels of wol_shell_traj.hip, integer accumulators per (frame, population) segment on the device.
// ---------------------------------------------------------------------------------------------
struct ShellTrajOut {
    int kind = 0;                 // 0 = three-body angles, 1 = LSI
    int nbins = 0;
    double hist_max = 0.0;
    int64_t* hist = nullptr;      // kind 0: [T][npops + 1][nbins] written; kind 1: [npops + 1][nbins] accumulated
    int64_t* counts = nullptr;    // kind 0: [T][npops + 1][2]; kind 1: [T][npops + 1]
    double* a = nullptr;          // kind 0: cos sums [T][npops + 1][2]; kind 1: mean [T][npops + 1]
    double* b = nullptr;          // kind 1: var
    void* per_water = nullptr;    // kind 1: [T][n] reals of the call's dtype or null
    void* per_point = nullptr;    // kind 1: [total points] or null
};

template <typename T>
int shell_traj_impl(wol_ctx* c, const T* frames, int64_t atom_stride, int64_t frame_stride, int T_frames, int n, const double* boxes,
                    double low, double high, const PopQuery& pq, const ShellTrajOut& o, int mem) {
    WOL_TRY(check_ctx(c));
    if (!frames || !boxes) return fail(WOL_ERR_INVALID, "frames/boxes must not be NULL");
    if (T_frames < 0 || n <= 0) return fail(WOL_ERR_INVALID, "need T >= 0 and n > 0 (got T=%d n=%d)", T_frames, n);
    if (atom_stride < 3) return fail(WOL_ERR_INVALID, "atom_stride must be >= 3");
    if (frame_stride < (int64_t)(n - 1) * atom_stride + 3 && T_frames > 1) return fail(WOL_ERR_INVALID, "frame_stride smaller than one frame");
    if (mem != WOL_MEM_HOST && mem != WOL_MEM_DEVICE) return fail(WOL_ERR_INVALID, "bad mem space");
    if (o.nbins <= 0 || o.nbins > 16000) return fail(WOL_ERR_INVALID, "need 0 < nbins <= 16000");
    if (o.kind == 1 && !(o.hist_max > 0.0)) return fail(WOL_ERR_INVALID, "histogram range must be positive");
    if (!(high > 0.0) || !(low >= 0.0)) return fail(WOL_ERR_INVALID, "need lowcut >= 0 and highcut > 0");
    const int npops = pq.npops;
    if (npops < 0) return fail(WOL_ERR_INVALID, "npops must be >= 0");
    if (T_frames == 0) return WOL_OK;
    const bool host = (mem == WOL_MEM_HOST);
    const int dtype = std::is_same<T, double>::value ? WOL_F64 : WOL_F32;
    const int npops1 = npops + 1, nseg = T_frames * npops1;

    FrameFeeder<T> feed;
    WOL_TRY(feed.plan(c, frames, atom_stride, frame_stride, T_frames, n, host, 96.0));
    const int nbatch = feed.nbatch;
    const double* d_pts = nullptr;
    const long long* d_poff = nullptr;
    int64_t npts = 0;
    WOL_TRY(stage_populations(c, pq, T_frames, dtype, mem, &d_pts, &d_poff, &npts));

    const size_t stat_b = sizeof(unsigned long long) * 8 * (size_t)nseg;
    const size_t hist_n = (o.kind == 0 ? (size_t)nseg : (size_t)npops1) * o.nbins;
    WOL_TRY(c->pop_stat.ensure(stat_b));
    WOL_TRY(c->pop_hist.ensure(sizeof(unsigned long long) * hist_n));
    HIP_TRY(hipMemsetAsync(c->pop_stat.p, 0, stat_b, c->stream));
    HIP_TRY(hipMemsetAsync(c->pop_hist.p, 0, sizeof(unsigned long long) * hist_n, c->stream));
    // waters whose shell outgrows the kernels' LDS rows are listed and finished by the slow kernels: the three-body one
    // recomputes, the LSI one keeps one row of WOL_TRAJ_LSI_SLOWROW doubles per lane of its grid in global memory: no storage
    // per listed water, so every water (and, for the LSI, every population point) of a batch may be listed
    const int64_t batch_waters = (int64_t)feed.Fb * n;
    int64_t batch_pts = 0;
    if (o.kind == 1 && npops > 0)
        for (int b = 0; b < nbatch; ++b)
            batch_pts = std::max<int64_t>(batch_pts, pq.off[(size_t)feed.bstart[b + 1] * npops] - pq.off[(size_t)feed.bstart[b] * npops]);
    const int slow_cap = (int)std::min<int64_t>(batch_waters + batch_pts, 1 << 30);
    const int slow_grid = 16;
    WOL_TRY(c->slow_list.ensure(sizeof(unsigned long long) * (size_t)slow_cap));
    WOL_TRY(c->slow_count.ensure(sizeof(int) * 4));
    HIP_TRY(hipMemsetAsync(c->slow_count.p, 0, sizeof(int) * 4, c->stream));
    if (o.kind == 1) WOL_TRY(c->qbuf.ensure(sizeof(double) * (size_t)slow_grid * WOL_TRAJ_BLOCK * WOL_TRAJ_LSI_SLOWROW));
    if (o.per_point && npts > 0) WOL_TRY(c->pop_out.ensure(sizeof(double) * (size_t)npts));
    const bool pw_direct = o.per_water && !host && dtype == WOL_F64;
    // cells as wide as the reach (27-cell walk) for short reaches, half the reach (125 cells) for long ones
    const double reach = o.kind == 0 ? high : high + 3.7;     // water_properties.py:276, 281
    const double cell_w = reach < 5.0 ? reach * 1.001 : std::fmax(reach * 0.5, 2.5);
    // LSI: LDS rows for twice the shell the first frame's mean density predicts (+ the next neighbour + head-room)
    int lsi_row = WOL_TRAJ_LSI_ROW;
    if (o.kind == 1 && boxes[0] > 0.0 && boxes[1] > 0.0 && boxes[2] > 0.0) {
        const double rho = (double)n / (boxes[0] * boxes[1] * boxes[2]);
        const double kexp = rho * (4.0 / 3.0) * WOL_PI * (high * high * high - low * low * low);
        lsi_row = (int)std::fmin((double)WOL_TRAJ_LSI_ROWMAX, std::fmax((double)WOL_TRAJ_LSI_ROW, 1.6 * kexp + 5.0));
    }
    bool lsi_lds_set = false;
    WOL_TRY(feed.start());
    for (int b = 0; b < nbatch; ++b) {
        const int f0 = feed.bstart[b], F = feed.bstart[b + 1] - f0;
        FrameBatchView bv;
        WOL_TRY(feed.acquire(b, &bv));
        BatchPlan bp;
        if (bv.f32) WOL_TRY(sort_batch<float>(c, view_as<float>(bv, n), F, boxes + 3 * (size_t)f0, cell_w, reach, false, false, 0, &bp, true));
        else WOL_TRY(sort_batch<double>(c, view_as<double>(bv, n), F, boxes + 3 * (size_t)f0, cell_w, reach, false, false, 0, &bp, true));
        wol_ctx::SortSet& S = *bp.S;
        TrajShellArgs ta;
        memset(&ta, 0, sizeof ta);
        ta.grids = S.grids.as<FrameGrid>(); ta.sorted = S.sorted.as<double4>(); ta.cell_start = S.cells.as<int>();
        ta.n = n; ta.F = F; ta.f0 = f0; ta.npops = npops;
        ta.lowsq = low * low; ta.highsq = high * high;
        ta.reach = reach;
        ta.pts = d_pts; ta.pt_off = d_poff;
        ta.nbins = o.nbins; ta.hist_max = o.hist_max;
        ta.hist = c->pop_hist.as<unsigned long long>(); ta.stat = c->pop_stat.as<unsigned long long>();
        ta.slow_list = c->slow_list.as<unsigned long long>();
        ta.slow_count = c->slow_count.as<int>(); ta.error_flag = c->slow_count.as<int>() + 1; ta.slow_cap = slow_cap;
        HIP_TRY(hipMemsetAsync(c->slow_count.p, 0, sizeof(int), c->stream));   // (the error flag stays up)
        double* pw_stage = nullptr;
        if (o.per_water) {
            if (pw_direct) ta.per_water = static_cast<double*>(o.per_water);
            else {
                WOL_TRY(c->out_stage.ensure(sizeof(double) * (size_t)feed.Fb * n));
                pw_stage = c->out_stage.as<double>();
                ta.per_water = pw_stage - (int64_t)f0 * n;    // (indexed by the frame's number in the call)
            }
        }
        const int64_t pts_here = npops > 0 ? pq.off[(size_t)(f0 + F) * npops] - pq.off[(size_t)f0 * npops] : 0;
        const size_t lds = sizeof(int) * (size_t)o.nbins;
        const int64_t need = (int64_t)F * ((n + WOL_TRAJ_BLOCK - 1) / WOL_TRAJ_BLOCK);
        int grid = 0;
        std::pair<hipEvent_t, hipEvent_t> ev;
        if (c->time_kernels) WOL_TRY(begin_timing(c, &ev));
        if (o.kind == 0) {
            if (lds > 16 * 1024 && !c->cos_traj_lds_set) {   // (the kernel's own arrays take 40 KB: beyond 64 KB in all, say so)
                HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_cos_angs_traj), hipFuncAttributeMaxDynamicSharedMemorySize, 112 * 1024));
                c->cos_traj_lds_set = true;
            }
            WOL_TRY(resident_grid(c, k_cos_angs_traj, WOL_TRAJ_BLOCK, lds, need, &grid));
            hipLaunchKernelGGL(k_cos_angs_traj, dim3(grid), dim3(WOL_TRAJ_BLOCK), lds, c->stream, ta);
            if (c->time_kernels) WOL_TRY(end_timing(c, ev));
            hipLaunchKernelGGL(k_cos_angs_traj_slow, dim3(slow_grid), dim3(WOL_TRAJ_BLOCK), 0, c->stream, ta);
            if (pts_here > 0)
                hipLaunchKernelGGL(k_cos_angs_traj_points, dim3((unsigned)std::min<int64_t>((pts_here + WOL_TRAJ_BLOCK - 1) / WOL_TRAJ_BLOCK, 4096)),
                                   dim3(WOL_TRAJ_BLOCK), 0, c->stream, ta);
        } else {
            const double high2 = high + 3.7;
            const size_t rows_lds = sizeof(double) * (size_t)lsi_row * WOL_TRAJ_BLOCK;
            if (!lsi_lds_set && rows_lds + lds > 48 * 1024) {
                HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_lsi_traj), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 2048));
                HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_lsi_traj_points), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 2048));
                lsi_lds_set = true;
            }
            WOL_TRY(resident_grid(c, k_lsi_traj, WOL_TRAJ_BLOCK, rows_lds + lds, need, &grid));
            hipLaunchKernelGGL(k_lsi_traj, dim3(grid), dim3(WOL_TRAJ_BLOCK), rows_lds + lds, c->stream, ta, high2 * high2, lsi_row);
            if (c->time_kernels) WOL_TRY(end_timing(c, ev));
            double* ptq = (o.per_point && npts > 0) ? c->pop_out.as<double>() : nullptr;
            if (pts_here > 0)
                hipLaunchKernelGGL(k_lsi_traj_points, dim3((unsigned)std::min<int64_t>((pts_here + WOL_TRAJ_BLOCK - 1) / WOL_TRAJ_BLOCK, 4096)),
                                   dim3(WOL_TRAJ_BLOCK), rows_lds, c->stream, ta, high2 * high2, lsi_row, ptq);
            hipLaunchKernelGGL(k_lsi_traj_slow, dim3(slow_grid), dim3(WOL_TRAJ_BLOCK), 0, c->stream, ta, high2 * high2, c->qbuf.as<double>(), ptq);
        }
        HIP_TRY(hipEventRecord(S.free_ev, c->stream));
        S.used = true;
        if (pw_stage)
            WOL_TRY(deliver_f64(c, pw_stage, static_cast<char*>(o.per_water) + (size_t)f0 * n * (dtype == WOL_F64 ? sizeof(double) : sizeof(float)),
                                (size_t)F * n, dtype, mem, 0));
        WOL_TRY(feed.release(b));
    }
    // ---- per-segment results from the integer accumulators ----
    const hipMemcpyKind back = host ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice;
    const size_t per_seg = o.kind == 0 ? 2 : 1;
    WOL_TRY(c->pop_res.ensure((sizeof(long long) + 2 * sizeof(double)) * per_seg * (size_t)nseg + 64));
    long long* r_cnt = c->pop_res.as<long long>();
    double* r_a = reinterpret_cast<double*>(r_cnt + per_seg * (size_t)nseg);
    double* r_b = r_a + per_seg * (size_t)nseg;
    hipLaunchKernelGGL(k_traj_finalize, dim3((nseg + 255) / 256), dim3(256), 0, c->stream, c->pop_stat.as<unsigned long long>(), nseg, 0,
                       npops1, o.kind, 0, r_cnt, r_a, r_b);
    if (o.counts) HIP_TRY(hipMemcpyAsync(o.counts, r_cnt, sizeof(long long) * per_seg * (size_t)nseg, back, c->stream));
    if (o.a) HIP_TRY(hipMemcpyAsync(o.a, r_a, sizeof(double) * per_seg * (size_t)nseg, back, c->stream));
    if (o.kind == 1 && o.b) HIP_TRY(hipMemcpyAsync(o.b, r_b, sizeof(double) * (size_t)nseg, back, c->stream));
    if (o.per_point && npts > 0) WOL_TRY(deliver_f64(c, c->pop_out.as<double>(), o.per_point, (size_t)npts, dtype, mem, 3));
    if (o.hist) {
        if (o.kind == 0) {
            HIP_TRY(hipMemcpyAsync(o.hist, c->pop_hist.p, sizeof(long long) * hist_n, back, c->stream));
        } else if (!host) {
            hipLaunchKernelGGL(k_add_u64, dim3((unsigned)((hist_n + 255) / 256)), dim3(256), 0, c->stream, c->pop_hist.as<unsigned long long>(),
                               reinterpret_cast<unsigned long long*>(o.hist), (int)hist_n);
        }
    }
    // the shell kernels report a shell too dense for their rows through a flag: one synchronisation per call
    int flags[2] = {0, 0};
    HIP_TRY(hipMemcpyAsync(flags, c->slow_count.p, sizeof flags, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (o.hist && o.kind == 1 && host) {
        std::vector<unsigned long long> h(hist_n);
        HIP_TRY(hipMemcpy(h.data(), c->pop_hist.p, sizeof(unsigned long long) * hist_n, hipMemcpyDeviceToHost));
        for (size_t k = 0; k < hist_n; ++k) o.hist[k] += (int64_t)h[k];
    }
    HIP_TRY(hipGetLastError());
    if (flags[1] == 1) return fail(WOL_ERR_INVALID, "the list of dense shells overflowed (%d entries)", slow_cap);
    if (flags[1] == 2) return fail(WOL_ERR_INVALID, "a shell holds more than %d waters; use the per-frame call", WOL_TRAJ_LSI_SLOWROW - 2);
    return WOL_OK;
}

// ---------------------------------------------------------------------------------------------
// rdfCalc's frame loop for the one-set g(r) (orderParam_lib.py:624-630: wl.radialdistsame per frame): batches through the
// feeder and the sort stage (cells half the reach wide), one launch of the tiled pair kernel per frame, integer counts per
// frame left on the device; the reference's normalisation is redone on the host from the counts, frame by frame.
// ---------------------------------------------------------------------------------------------
// rdf(k) = counts(k) / (NPos * BulkDens * (4./3.) * pi * binwidth**3 * (k**3 - (k-1)**3)) with the reference's
// single-precision 4./3., truncated pi and integer k**3 (waterlib.f90:227-229, 353-355)
void rdf_normalise(const std::vector<long long>& counts, int npos, double bulkdens, double binwidth, double* rdf) {
    const double pi = 3.141592653589;
    const double fourthirds = (double)(4.f / 3.f);
    for (int k = 1; k <= (int)counts.size(); ++k) {
        const int shell = k * k * k - (k - 1) * (k - 1) * (k - 1);
        const double den = (double)npos * bulkdens * fourthirds * pi * (binwidth * binwidth * binwidth) * (double)shell;
        rdf[k - 1] = (double)counts[(size_t)k - 1] / den;
    }
}


template <typename T>
int pair_hist_traj_impl(wol_ctx* c, const T* frames, int64_t atom_stride, int64_t frame_stride, int T_frames, int n, const double* boxes,
                        double binwidth, int totbins, double bulkdens, double* rdf, int64_t* counts, int mem) {
    WOL_TRY(check_ctx(c));
    if (!frames || !boxes) return fail(WOL_ERR_INVALID, "frames/boxes must not be NULL");
    if (T_frames < 0 || n <= 0) return fail(WOL_ERR_INVALID, "need T >= 0 and n > 0 (got T=%d n=%d)", T_frames, n);
    if (atom_stride < 3) return fail(WOL_ERR_INVALID, "atom_stride must be >= 3");
    if (frame_stride < (int64_t)(n - 1) * atom_stride + 3 && T_frames > 1) return fail(WOL_ERR_INVALID, "frame_stride smaller than one frame");
    if (mem != WOL_MEM_HOST && mem != WOL_MEM_DEVICE) return fail(WOL_ERR_INVALID, "bad mem space");
    if (!(binwidth > 0.0) || totbins <= 0 || totbins > 4000) return fail(WOL_ERR_INVALID, "rdf: need binwidth > 0 and 0 < totbins <= 4000");
    if (rdf && totbins > 1290) return fail(WOL_ERR_INVALID, "rdf: totbins > 1290 overflows the reference's integer k**3");
    if (T_frames == 0) return WOL_OK;
    const bool host = (mem == WOL_MEM_HOST);
    const double reach = binwidth * (double)totbins;
    const double cell_w = reach < 5.0 ? reach * 1.001 : std::fmax(reach * 0.5, 2.5);
    FrameFeeder<T> feed;
    WOL_TRY(feed.plan(c, frames, atom_stride, frame_stride, T_frames, n, host, 96.0));
    const size_t hb = sizeof(unsigned long long) * (size_t)T_frames * totbins;
    WOL_TRY(c->pop_hist.ensure(hb));
    HIP_TRY(hipMemsetAsync(c->pop_hist.p, 0, hb, c->stream));
    const size_t lds = sizeof(int) * (size_t)totbins;
    WOL_TRY(feed.start());
    for (int b = 0; b < feed.nbatch; ++b) {
        const int f0 = feed.bstart[b], F = feed.bstart[b + 1] - f0;
        FrameBatchView bv;
        WOL_TRY(feed.acquire(b, &bv));
        BatchPlan bp;
        if (bv.f32) WOL_TRY(sort_batch<float>(c, view_as<float>(bv, n), F, boxes + 3 * (size_t)f0, cell_w, reach, false, false, 0, &bp, true));
        else WOL_TRY(sort_batch<double>(c, view_as<double>(bv, n), F, boxes + 3 * (size_t)f0, cell_w, reach, false, false, 0, &bp, true));
        wol_ctx::SortSet& S = *bp.S;
        std::pair<hipEvent_t, hipEvent_t> ev;
        if (c->time_kernels) WOL_TRY(begin_timing(c, &ev));
        for (int f = 0; f < F; ++f) {
            const FrameGrid* gh = bp.gh + f;
            unsigned long long* hist = c->pop_hist.as<unsigned long long>() + (size_t)(f0 + f) * totbins;
            int grid = 0;
            if (gh->maxshell >= 0 && c->pair_tiled && reach < 60.0) {
                PairTileArgs pa;
                pa.grid = S.grids.as<FrameGrid>() + f; pa.sorted = S.sorted.as<double4>() + (size_t)f * n; pa.cell_start = S.cells.as<int>(); pa.n = n;
                const double wpc = std::fmax((double)n / (double)gh->ncell, 0.05);
                int cxrun = std::min((int)std::fmax(1.0, std::floor(58.0 / wpc)), gh->nc[0]);
                pa.ntx = (gh->nc[0] + cxrun - 1) / cxrun;
                pa.cxrun = (gh->nc[0] + pa.ntx - 1) / pa.ntx;
                pa.need_minimg = 0;
                const double ext[3] = {0.5 * pa.cxrun * gh->cw[0], 0.5 * gh->cw[1], 0.5 * gh->cw[2]};
                for (int k = 0; k < 3; ++k)
                    if (gh->box.L[k] < 2.0 * (ext[k] + reach) + 1.0) pa.need_minimg = 1;
                pa.binwidth = binwidth; pa.totbins = totbins;
                pa.cut2f = (float)(reach * reach * (1.0 + 1e-5) + 1e-3);
                pa.hist = hist;
                const int64_t units = (int64_t)gh->nc[1] * gh->nc[2] * pa.ntx * (gh->lo[1] + gh->hi[1] + 1) * (gh->lo[2] + gh->hi[2] + 1);
                if (pa.need_minimg) {
                    WOL_TRY(resident_grid(c, k_pair_hist_tiled<true>, WOL_PT_BLOCK, lds, units, &grid));
                    hipLaunchKernelGGL(k_pair_hist_tiled<true>, dim3(grid), dim3(WOL_PT_BLOCK), lds, c->stream, pa);
                } else {
                    WOL_TRY(resident_grid(c, k_pair_hist_tiled<false>, WOL_PT_BLOCK, lds, units, &grid));
                    hipLaunchKernelGGL(k_pair_hist_tiled<false>, dim3(grid), dim3(WOL_PT_BLOCK), lds, c->stream, pa);
                }
            } else {
                ShellArgs sa;
                sa.grid = S.grids.as<FrameGrid>() + f; sa.sorted = S.sorted.as<double4>() + (size_t)f * n; sa.cell_start = S.cells.as<int>();
                sa.n = n; sa.sub = nullptr; sa.ns = n; sa.use_cells = gh->maxshell >= 0 ? 1 : 0; sa.lowsq = 0.0; sa.highsq = 0.0;
                WOL_TRY(resident_grid(c, k_pair_hist, WOL_BLOCK, lds, (n + WOL_BLOCK - 1) / WOL_BLOCK, &grid));
                hipLaunchKernelGGL(k_pair_hist, dim3(grid), dim3(WOL_BLOCK), lds, c->stream, sa, 1, binwidth, totbins, hist);
            }
        }
        if (c->time_kernels) WOL_TRY(end_timing(c, ev));
        HIP_TRY(hipEventRecord(S.free_ev, c->stream));
        S.used = true;
        WOL_TRY(feed.release(b));
    }
    std::vector<long long> hc((size_t)T_frames * totbins);
    HIP_TRY(hipMemcpyAsync(hc.data(), c->pop_hist.p, hb, hipMemcpyDeviceToHost, c->stream));
    if (counts && !host) HIP_TRY(hipMemcpyAsync(counts, c->pop_hist.p, hb, hipMemcpyDeviceToDevice, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(hipGetLastError());
    if (counts && host) memcpy(counts, hc.data(), hb);
    if (rdf) {
        std::vector<double> r((size_t)T_frames * totbins);
        std::vector<long long> one((size_t)totbins);
        for (int t = 0; t < T_frames; ++t) {
            std::copy(hc.begin() + (size_t)t * totbins, hc.begin() + (size_t)(t + 1) * totbins, one.begin());
            rdf_normalise(one, n, bulkdens, binwidth, r.data() + (size_t)t * totbins);
        }
        if (host) memcpy(rdf, r.data(), sizeof(double) * r.size());
        else HIP_TRY(hipMemcpy(rdf, r.data(), sizeof(double) * r.size(), hipMemcpyHostToDevice));
    }
    return WOL_OK;
}

}  // namespace

// =============================================================================================
// C ABI
// =============================================================================================
extern "C" {

int wol_version(void) { return 100; }

const char* wol_last_error(void) { return g_err.c_str(); }

int wol_ctx_create(int device, wol_ctx** out) {
    if (!out) return fail(WOL_ERR_INVALID, "out must not be NULL");
    *out = nullptr;
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0) return fail(WOL_ERR_NODEVICE, "no HIP device available (%s)", hipGetErrorString(e));
    if (device < 0 || device >= count) return fail(WOL_ERR_INVALID, "device %d out of range (have %d)", device, count);
    HIP_TRY(hipSetDevice(device));
    wol_ctx* c = new (std::nothrow) wol_ctx();
    if (!c) return fail(WOL_ERR_NOMEM, "out of host memory");
    c->device = device;
    {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0)
            c->num_cus = prop.multiProcessorCount;
    }
    hipError_t se = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (se != hipSuccess) { delete c; return fail(WOL_ERR_HIP, "hipStreamCreate: %s", hipGetErrorString(se)); }
    c->own_stream = true;
    *out = c;
    return WOL_OK;
}

int wol_ctx_destroy(wol_ctx* c) {
    if (!c) return WOL_OK;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    for (auto& ev : c->ev_used) { (void)hipEventDestroy(ev.first); (void)hipEventDestroy(ev.second); }
    for (auto& ev : c->ev_free) { (void)hipEventDestroy(ev.first); (void)hipEventDestroy(ev.second); }
    DevBuf* bufs[] = {&c->cellid, &c->rank, &c->qbuf, &c->in_stage,
                      &c->out_stage, &c->hist_stage, &c->small, &c->stat, &c->layer_off, &c->sort_tmp, &c->slow_list, &c->slow_count, &c->stamps, &c->scan_tot, &c->seg_tab};
    for (auto& S : c->sets) {
        S.sorted.release(); S.sorted32.release(); S.cells.release(); S.grids.release(); S.level.release(); S.cnt4.release(); S.prop.release(); S.prop_key.release(); S.work.release(); S.tile_any.release();
        if (S.rows_ev) (void)hipEventDestroy(S.rows_ev);
        if (S.level_ev) (void)hipEventDestroy(S.level_ev);
        if (S.searched_ev) (void)hipEventDestroy(S.searched_ev);
        if (S.sorted_ev) (void)hipEventDestroy(S.sorted_ev);
        if (S.free_ev) (void)hipEventDestroy(S.free_ev);
    }
    if (c->in_ev) (void)hipEventDestroy(c->in_ev);
    if (c->scratch_ev) (void)hipEventDestroy(c->scratch_ev);
    if (c->sort_stream) { (void)hipStreamSynchronize(c->sort_stream); (void)hipStreamDestroy(c->sort_stream); }
    if (c->search_stream) { (void)hipStreamSynchronize(c->search_stream); (void)hipStreamDestroy(c->search_stream); }
    if (c->grid_stream) { (void)hipStreamSynchronize(c->grid_stream); (void)hipStreamDestroy(c->grid_stream); }
    for (DevBuf* b : bufs) b->release();
    c->cvt_in.release(); c->xchg.release(); c->stat_part.release(); c->prop_counters.release();
    c->pop_off.release(); c->pop_pts.release(); c->pop_stat.release(); c->pop_hist.release(); c->pop_out.release(); c->pop_res.release();
    for (auto& b : c->cvt_out) b.release();
    for (int k = 0; k < 2; ++k) {
        c->host_stage[k].release();
        c->pack_stage[k].release();
        if (c->copied_ev[k]) (void)hipEventDestroy(c->copied_ev[k]);
        if (c->consumed_ev[k]) (void)hipEventDestroy(c->consumed_ev[k]);
    }
    if (c->copy_stream) { (void)hipStreamSynchronize(c->copy_stream); (void)hipStreamDestroy(c->copy_stream); }
    c->grids_host.release();
    if (c->grid_ev) (void)hipEventDestroy(c->grid_ev);
    if (c->shell_ev) (void)hipEventDestroy(c->shell_ev);
    if (c->own_stream) (void)hipStreamDestroy(c->stream);
    delete c;
    return WOL_OK;
}

int wol_ctx_set_stream(wol_ctx* c, void* hip_stream) {
    WOL_TRY(check_ctx(c));
    hipStream_t s = reinterpret_cast<hipStream_t>(hip_stream);  // NULL = HIP's default (null) stream
    if (s == c->stream && !c->own_stream) return WOL_OK;
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (c->own_stream) { (void)hipStreamDestroy(c->stream); c->own_stream = false; }
    c->stream = s;
    return WOL_OK;
}

int wol_ctx_synchronize(wol_ctx* c) {
    WOL_TRY(check_ctx(c));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return WOL_OK;
}

int wol_ctx_set_option(wol_ctx* c, const char* key, double value) {
    if (!c || !key) return fail(WOL_ERR_INVALID, "null argument");
    const std::string k(key);
    if (k == "cell_width") {
        if (!(value >= 0.5 && value <= 1e6)) return fail(WOL_ERR_INVALID, "cell_width out of range");
        c->cell_width = value;
    } else if (k == "frames_per_batch") {
        c->frames_per_batch = value < 0 ? 0 : (int)value;
    } else if (k == "path") {
        if (value < 0 || value > 3) return fail(WOL_ERR_INVALID, "path must be 0..3");
        c->path = (int)value;
    } else if (k == "inputs_resident") {
        c->inputs_resident = value != 0.0;
    } else if (k == "search_rows") {
        c->search_rows = value != 0.0;
    } else if (k == "rows_block") {
        const int v = (int)value;
        if (v != 0 && v != WOL_ROWS_BLOCK_A && v != WOL_ROWS_BLOCK_B)
            return fail(WOL_ERR_INVALID, "rows_block must be 0, %d or %d", WOL_ROWS_BLOCK_A, WOL_ROWS_BLOCK_B);
        c->force_rows_block = v;
    } else if (k == "sort_parts") {
        if (!(value >= 0 && value <= 64)) return fail(WOL_ERR_INVALID, "sort_parts must be 0..64");
        c->sort_parts = (int)value;
    } else if (k == "dyn_tiles") {
        if (value != 0.0 && value != 1.0) return fail(WOL_ERR_INVALID, "dyn_tiles must be 0 or 1");
        c->dyn_tiles = (int)value;
    } else if (k == "seg_sort") {
        if (value != 0.0 && value != 1.0 && value != 2.0) return fail(WOL_ERR_INVALID, "seg_sort must be 0, 1 or 2");
        c->seg_sort = (int)value;
    } else if (k == "stages") {
        if (value != 2.0 && value != 3.0) return fail(WOL_ERR_INVALID, "stages must be 2 or 3");
        c->stages = (int)value;
    } else if (k == "search_margin") {
        if (!(value >= 0 && value <= 2)) return fail(WOL_ERR_INVALID, "search_margin must be 0..2");
        c->search_margin = (int)value;
    } else if (k == "overlap") {
        c->overlap = value != 0.0;
    } else if (k == "skip_empty_tiles") {
        c->skip_empty = value != 0.0;
    } else if (k == "pair_tiled") {
        c->pair_tiled = value != 0.0;
    } else if (k == "rec32") {
        c->rec32 = value != 0.0 ? 1 : 0;
    } else if (k == "idx_sort") {
        c->idx_sort = value != 0.0 ? 1 : 0;
    } else if (k == "host_pack") {
        if (value != 0.0 && value != 1.0 && value != 2.0) return fail(WOL_ERR_INVALID, "host_pack must be 0, 1 or 2");
        c->host_pack = (int)value;
    } else if (k == "host_threads") {
        if (!(value >= 0 && value <= 256)) return fail(WOL_ERR_INVALID, "host_threads must be 0..256");
        c->host_threads = (int)value;
    } else if (k == "check_proposals") {
        // (re)starts the counters "proposals_written" / "proposals_expected" / "proposals_bad"
        WOL_TRY(check_ctx(c));
        WOL_TRY(c->prop_counters.ensure(sizeof(unsigned long long) * 2));
        HIP_TRY(hipStreamSynchronize(c->stream));
        HIP_TRY(hipMemset(c->prop_counters.p, 0, sizeof(unsigned long long) * 2));
        c->prop_expected = 0;
        c->check_props = value != 0.0;
    } else if (k == "lds_sort") {
        c->lds_sort = value != 0.0;
    } else if (k == "time_kernels") {
        c->time_kernels = value != 0.0;
    } else {
        return fail(WOL_ERR_INVALID, "unknown option '%s'", key);
    }
    return WOL_OK;
}

int wol_ctx_kernel_time(wol_ctx* c, double* ms_total, int64_t* launches, int reset) {
    WOL_TRY(check_ctx(c));
    WOL_TRY(collect_timing(c));
    if (ms_total) *ms_total = c->ms_total;
    if (launches) *launches = c->launches;
    if (reset) { c->ms_total = 0.0; c->launches = 0; }
    return WOL_OK;
}

int wol_ctx_get_counter(wol_ctx* c, const char* key, int64_t* value) {
    WOL_TRY(check_ctx(c));
    if (!key || !value) return fail(WOL_ERR_INVALID, "null argument");
    if (std::string(key) == "slow_waters") {
        *value = 0;
        if (c->last_set && c->last_set->work.p) {   // length of the last batch's exact list
            int v = 0;
            HIP_TRY(hipStreamSynchronize(c->stream));
            HIP_TRY(hipMemcpy(&v, c->last_set->work.as<int>() + 16 * 8, sizeof(int), hipMemcpyDeviceToHost));
            *value = v;
        }
        return WOL_OK;
    }
    if (std::string(key) == "proposals_written" || std::string(key) == "proposals_bad" || std::string(key) == "proposals_expected") {
        // diagnostic counters of option "check_proposals": keys the row-merged proposal kernel wrote / proposals naming a
        // slot outside their frame / waters it was given, since the option was switched on
        if (!c->prop_counters.p) return fail(WOL_ERR_INVALID, "counter '%s' needs option check_proposals", key);
        if (key[10] == 'e') { *value = c->prop_expected; return WOL_OK; }
        unsigned long long v[2] = {0, 0};
        HIP_TRY(hipStreamSynchronize(c->stream));
        if (c->search_stream) HIP_TRY(hipStreamSynchronize(c->search_stream));
        HIP_TRY(hipMemcpy(v, c->prop_counters.p, sizeof v, hipMemcpyDeviceToHost));
        *value = (int64_t)v[key[10] == 'w' ? 0 : 1];
        return WOL_OK;
    }
    if (std::string(key) == "h2d_bytes") {   // bytes the last host-memory trajectory call sent over the link
        *value = (int64_t)c->last_h2d_bytes;
        return WOL_OK;
    }
    if (std::string(key) == "cells_x" || std::string(key) == "cells_y" || std::string(key) == "cells_z") {   // grid of frame 0 of the last batch
        const FrameGrid* g = c->grids_host.as<FrameGrid>();
        *value = g ? g->nc[key[6] - 'x'] : 0;
        return WOL_OK;
    }
    if (std::string(key) == "tile_sx" || std::string(key) == "tile_ry" || std::string(key) == "tiles_per_frame") {
        // geometry of the proposal kernel's tiles for frame 0 of the last batch at density level 0 (diagnostic)
        // ... at the density level the device picked for it
        const FrameGrid* g = c->grids_host.as<FrameGrid>();
        int lv = 0;
        if (g && c->last_set && c->last_set->level.p) {
            HIP_TRY(hipStreamSynchronize(c->stream));
            HIP_TRY(hipMemcpy(&lv, c->last_set->level.p, sizeof(int), hipMemcpyDeviceToHost));
            lv &= 0xFF;
            if (lv < 0 || lv >= WOL_NLEV) lv = 0;
        }
        *value = !g ? 0 : key[5] == 's' && key[4] == '_' ? (c->search_rows ? g->row_sx[lv] : g->lev_sx[lv])
                          : key[5] == 'r' ? (c->search_rows ? g->row_ry[lv] : g->lev_ry[lv])
                          : (c->search_rows ? g->row_ntiles[lv] : g->lev_ntiles[lv]);
        return WOL_OK;
    }
    if (std::string(key) == "rows_block") {  // k_search_rows instantiation of the last batch (threads per block)
        *value = c->last_rows_block;
        return WOL_OK;
    }
    if (std::string(key) == "record_bytes") {  // storage form of the last q batch's sorted records (wol_rec): 32, or 16 for float32 frames
        *value = c->last_rec_bytes;
        return WOL_OK;
    }
    if (std::string(key) == "density_level") {  // tile-shape level of frame 0 of the last batch
        *value = -1;
        if (c->last_set && c->last_set->level.p) {
            int v = 0;
            HIP_TRY(hipStreamSynchronize(c->stream));
            HIP_TRY(hipMemcpy(&v, c->last_set->level.p, sizeof(int), hipMemcpyDeviceToHost));
            *value = v & 0xFF;    // (bit 8: the frame has tile rows without any water)
        }
        return WOL_OK;
    }
#ifdef WOL_PHASE_STAMPS
    if (std::string(key).rfind("stamp", 0) == 0 && c->stamps.p) {
        unsigned long long v[8];
        HIP_TRY(hipStreamSynchronize(c->stream));
        HIP_TRY(hipMemcpy(v, c->stamps.p, sizeof v, hipMemcpyDeviceToHost));
        *value = (int64_t)v[key[5] - '0'];
        if (key[6] == 'z') HIP_TRY(hipMemset(c->stamps.p, 0, sizeof v));
        return WOL_OK;
    }
#endif
    return fail(WOL_ERR_INVALID, "unknown counter '%s'", key);
}

// ---- level 1 ------------------------------------------------------------------------------
int wol_reimage(wol_ctx* c, const double* pos, int n, const double* refpos, const double* box, double* out) {
    WOL_TRY(check_ctx(c));
    if (!pos || !refpos || !box || !out || n < 0) return fail(WOL_ERR_INVALID, "reimage: bad argument");
    if (n == 0) return WOL_OK;
    const size_t bytes = sizeof(double) * 3 * (size_t)n;
    WOL_TRY(c->in_stage.ensure(bytes));
    WOL_TRY(c->out_stage.ensure(bytes));
    HIP_TRY(hipMemcpyAsync(c->in_stage.p, pos, bytes, hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(k_reimage, dim3((n + 255) / 256), dim3(256), 0, c->stream, c->in_stage.as<double>(), n,
                       refpos[0], refpos[1], refpos[2], make_box(box), c->out_stage.as<double>());
    HIP_TRY(hipMemcpyAsync(out, c->out_stage.p, bytes, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return WOL_OK;
}

static int mask_impl(wol_ctx* c, const double* sub, int ns, const double* pos, int n, const double* box,
                     double low, double high, int32_t* mask) {
    WOL_TRY(check_ctx(c));
    if (!pos || !sub || !box || !mask || n < 0 || ns < 0) return fail(WOL_ERR_INVALID, "nearneighbors: bad argument");
    if (n == 0 || ns == 0) return WOL_OK;
    const size_t pb = sizeof(double) * 3 * (size_t)n, sb = sizeof(double) * 3 * (size_t)ns;
    const size_t mb = sizeof(int32_t) * (size_t)n * ns;
    WOL_TRY(c->in_stage.ensure(pb + sb));
    WOL_TRY(c->out_stage.ensure(mb));
    double* dpos = c->in_stage.as<double>();
    double* dsub = dpos + 3 * (size_t)n;
    HIP_TRY(hipMemcpyAsync(dpos, pos, pb, hipMemcpyHostToDevice, c->stream));
    if (sub == pos) dsub = dpos;
    else HIP_TRY(hipMemcpyAsync(dsub, sub, sb, hipMemcpyHostToDevice, c->stream));
    PointsView<double> pv{dpos, 3, n}, sv{dsub, 3, ns};
    const dim3 grid((ns + WOL_BLOCK - 1) / WOL_BLOCK, (n + WOL_MASK_TJ - 1) / WOL_MASK_TJ);
    hipLaunchKernelGGL(k_brute_mask, grid, dim3(WOL_BLOCK), 0, c->stream, pv, sv, make_box(box), low * low,
                       high * high, c->out_stage.as<int32_t>());
    HIP_TRY(hipMemcpyAsync(mask, c->out_stage.p, mb, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return WOL_OK;
}

int wol_near_neighbors(wol_ctx* c, const double* sub, int ns, const double* pos, int n, const double* box,
                       double lowcut, double highcut, int32_t* mask) {
    return mask_impl(c, sub, ns, pos, n, box, lowcut, highcut, mask);
}

int wol_all_near_neighbors(wol_ctx* c, const double* pos, int n, const double* box, double lowcut,
                           double highcut, int32_t* mask) {
    return mask_impl(c, pos, n, pos, n, box, lowcut, highcut, mask);
}

int wol_tetra_cos_ang(wol_ctx* c, const double* refpos, const double* neigh, int n, const double* box,
                      double* allangs) {
    WOL_TRY(check_ctx(c));
    if (!refpos || !neigh || !box || !allangs || n < 0) return fail(WOL_ERR_INVALID, "tetracosang: bad argument");
    if (n == 0) return WOL_OK;
    const size_t ib = sizeof(double) * 3 * (size_t)n, ob = sizeof(double) * (size_t)n * n;
    WOL_TRY(c->in_stage.ensure(ib));
    WOL_TRY(c->out_stage.ensure(ob));
    HIP_TRY(hipMemcpyAsync(c->in_stage.p, neigh, ib, hipMemcpyHostToDevice, c->stream));
    const int64_t total = (int64_t)n * n;
    hipLaunchKernelGGL(k_tetra_cos_ang, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, c->stream,
                       c->in_stage.as<double>(), n, refpos[0], refpos[1], refpos[2], make_box(box),
                       c->out_stage.as<double>());
    HIP_TRY(hipMemcpyAsync(allangs, c->out_stage.p, ob, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return WOL_OK;
}

int wol_lsi_dists(wol_ctx* c, const double* refpos, const double* neigh, int n, const double* box, double* dist) {
    WOL_TRY(check_ctx(c));
    if (!refpos || !neigh || !box || !dist || n < 0) return fail(WOL_ERR_INVALID, "lsidists: bad argument");
    if (n == 0) return WOL_OK;
    const size_t ib = sizeof(double) * 3 * (size_t)n, ob = sizeof(double) * (size_t)n;
    WOL_TRY(c->in_stage.ensure(ib));
    WOL_TRY(c->out_stage.ensure(ob));
    HIP_TRY(hipMemcpyAsync(c->in_stage.p, neigh, ib, hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(k_lsi_dists, dim3((n + 255) / 256), dim3(256), 0, c->stream, c->in_stage.as<double>(), n,
                       refpos[0], refpos[1], refpos[2], make_box(box), c->out_stage.as<double>());
    HIP_TRY(hipMemcpyAsync(dist, c->out_stage.p, ob, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return WOL_OK;
}

// ---- level 2 ------------------------------------------------------------------------------
int wol_q_frame_t(wol_ctx* c, const void* pos, int64_t pos_stride, int n, const void* sub, int64_t sub_stride,
                  int ns, const double* box, double lowcut, double highcut, void* q, int32_t* nn_idx,
                  void* nn_dist, void* ang_deg, int dtype, int mem) {
    WOL_TRY(check_ctx(c));
    WOL_TRY(check_real(dtype, mem));
    if (!pos || !box || !q) return fail(WOL_ERR_INVALID, "q_frame: pos, box and q must not be NULL");
    if (n < 0 || ns < 0) return fail(WOL_ERR_INVALID, "q_frame: negative size");
    if (pos_stride < 3 || (sub && sub_stride < 3)) return fail(WOL_ERR_INVALID, "q_frame: stride must be >= 3");
    if (!sub) { ns = n; }
    if (ns == 0) return WOL_OK;
    const bool host = (mem == WOL_MEM_HOST);
    const bool direct = !host && dtype == WOL_F64;   // device doubles: kernels read and write the caller's arrays
    const size_t esz = dtype == WOL_F64 ? sizeof(double) : sizeof(float);
    QOut out;
    memset(&out, 0, sizeof out);
    FramesView<double> fv;
    fv.atom_stride = pos_stride;
    fv.frame_stride = 0;
    fv.n = n;
    const double* dsub = static_cast<const double*>(sub);
    if (n == 0) {
        // no candidates at all: every q is 0 (wp:358), padded outputs
        if (!host) return fail(WOL_ERR_INVALID, "q_frame: n == 0 with device buffers is not supported");
        memset(q, 0, esz * (size_t)ns);
        if (nn_idx) for (size_t k = 0; k < (size_t)ns * 4; ++k) nn_idx[k] = -1;
        for (size_t k = 0; k < (size_t)ns * 4 && nn_dist; ++k) {
            if (dtype == WOL_F64) static_cast<double*>(nn_dist)[k] = NAN; else static_cast<float*>(nn_dist)[k] = NAN;
        }
        for (size_t k = 0; k < (size_t)ns * 6 && ang_deg; ++k) {
            if (dtype == WOL_F64) static_cast<double*>(ang_deg)[k] = NAN; else static_cast<float*>(ang_deg)[k] = NAN;
        }
        return WOL_OK;
    }
    size_t o_dist = 0, o_ang = 0, o_idx = 0, o_total = 0;
    if (direct) {
        fv.base = static_cast<const double*>(pos);
        out.q = static_cast<double*>(q); out.nn_idx = nn_idx;
        out.nn_dist = static_cast<double*>(nn_dist); out.ang = static_cast<double*>(ang_deg);
    } else {
        const size_t pe = (size_t)(n - 1) * pos_stride + 3;
        const size_t se = sub ? (size_t)(ns - 1) * sub_stride + 3 : 0;
        WOL_TRY(c->in_stage.ensure(sizeof(double) * (pe + se)));
        WOL_TRY(fetch_f64(c, RealSrc{pos, dtype, mem}, pe, c->in_stage.as<double>()));
        fv.base = c->in_stage.as<double>();
        if (sub) {
            WOL_TRY(fetch_f64(c, RealSrc{sub, dtype, mem}, se, c->in_stage.as<double>() + pe));
            dsub = c->in_stage.as<double>() + pe;
        }
        o_total = sizeof(double) * (size_t)ns;
        if (nn_dist) { o_dist = o_total; o_total += sizeof(double) * 4 * (size_t)ns; }
        if (ang_deg) { o_ang = o_total; o_total += sizeof(double) * 6 * (size_t)ns; }
        if (nn_idx && host) { o_idx = o_total; o_total += sizeof(int32_t) * 4 * (size_t)ns; }
        WOL_TRY(c->out_stage.ensure(o_total));
        char* ob = c->out_stage.as<char>();
        out.q = reinterpret_cast<double*>(ob);
        out.nn_dist = nn_dist ? reinterpret_cast<double*>(ob + o_dist) : nullptr;
        out.ang = ang_deg ? reinterpret_cast<double*>(ob + o_ang) : nullptr;
        out.nn_idx = !nn_idx ? nullptr : host ? reinterpret_cast<int32_t*>(ob + o_idx) : nn_idx;
    }
    WOL_TRY(run_q_batch<double>(c, fv, 1, box, dsub, sub_stride, ns, lowcut, highcut, out));
    if (!direct) {
        WOL_TRY(deliver_f64(c, out.q, q, (size_t)ns, dtype, mem, 0));
        WOL_TRY(deliver_f64(c, out.nn_dist, nn_dist, 4 * (size_t)ns, dtype, mem, 1));
        WOL_TRY(deliver_f64(c, out.ang, ang_deg, 6 * (size_t)ns, dtype, mem, 2));
        if (nn_idx && host) HIP_TRY(hipMemcpyAsync(nn_idx, out.nn_idx, sizeof(int32_t) * 4 * (size_t)ns, hipMemcpyDeviceToHost, c->stream));
        if (host) HIP_TRY(hipStreamSynchronize(c->stream));
    }
    return WOL_OK;
}

int wol_q_frame(wol_ctx* c, const double* pos, int64_t pos_stride, int n, const double* sub, int64_t sub_stride,
                int ns, const double* box, double lowcut, double highcut, double* q, int32_t* nn_idx,
                double* nn_dist, double* ang_deg, int mem) {
    return wol_q_frame_t(c, pos, pos_stride, n, sub, sub_stride, ns, box, lowcut, highcut, q, nn_idx, nn_dist, ang_deg, WOL_F64, mem);
}

int wol_cos_angs_t(wol_ctx* c, const void* pos, int n, const void* sub, int ns, const double* box,
                   double lowcut, double highcut, void* ang, void* num_angs, int64_t* total, int dtype, int mem) {
    WOL_TRY(check_ctx(c));
    WOL_TRY(check_real(dtype, mem));
    if (!pos || !box || !total || n < 0 || ns < 0) return fail(WOL_ERR_INVALID, "cos_angs: bad argument");
    if (!sub) ns = n;
    *total = 0;
    if (ns == 0) return WOL_OK;
    const size_t esz = dtype == WOL_F64 ? sizeof(double) : sizeof(float);
    if (n == 0) {
        if (num_angs) {
            if (mem == WOL_MEM_HOST) memset(num_angs, 0, esz * (size_t)ns);
            else HIP_TRY(hipMemsetAsync(num_angs, 0, esz * (size_t)ns, c->stream));
        }
        return WOL_OK;
    }
    ShellArgs sa;
    WOL_TRY(shell_setup(c, RealSrc{pos, dtype, mem}, n, RealSrc{sub, dtype, mem}, ns, box, highcut, &sa));
    sa.lowsq = lowcut * lowcut;
    sa.highsq = highcut * highcut;
    // layout of `small`: count[ns] ints | off_nbr[ns] | off_ang[ns] | totals[2] (long long)
    const size_t o_off = ((sizeof(int) * (size_t)ns + 15) / 16) * 16;
    WOL_TRY(c->small.ensure(o_off + sizeof(long long) * (2 * (size_t)ns + 2)));
    int* d_count = c->small.as<int>();
    long long* d_offn = reinterpret_cast<long long*>(c->small.as<char>() + o_off);
    long long* d_offa = d_offn + ns;
    long long* d_tot = d_offa + ns;
    const dim3 grid((ns + WOL_BLOCK - 1) / WOL_BLOCK);
    hipLaunchKernelGGL(k_shell_count, grid, dim3(WOL_BLOCK), 0, c->stream, sa, d_count);
    hipLaunchKernelGGL(k_shell_offsets, dim3(1), dim3(1024), 0, c->stream, d_count, ns, d_offn, d_offa, d_tot);
    long long tot[2] = {0, 0};
    HIP_TRY(hipMemcpyAsync(tot, d_tot, sizeof tot, hipMemcpyDeviceToHost, c->stream));
    if (num_angs) {   // k per water as a real, like the reference's np.zeros array (wp:233)
        WOL_TRY(c->qbuf.ensure(sizeof(double) * (size_t)ns));
        hipLaunchKernelGGL(k_int_to_f64, grid, dim3(WOL_BLOCK), 0, c->stream, d_count, c->qbuf.as<double>(), (int64_t)ns);
        WOL_TRY(deliver_f64(c, c->qbuf.as<double>(), num_angs, (size_t)ns, dtype, mem, 0));
    }
    HIP_TRY(hipStreamSynchronize(c->stream));   // the totals size the second call's buffer: a host value either way
    *total = (int64_t)tot[1];
    if (!ang || tot[1] == 0) return WOL_OK;
    WOL_TRY(c->slow_list.ensure(sizeof(unsigned long long) * (size_t)std::max<long long>(tot[0], 1)));
    WOL_TRY(c->out_stage.ensure(sizeof(double) * (size_t)tot[1]));
    hipLaunchKernelGGL(k_cos_angs_fill, grid, dim3(WOL_BLOCK), 0, c->stream, sa, d_offn, d_offa,
                       c->slow_list.as<unsigned long long>(), c->out_stage.as<double>());
    WOL_TRY(deliver_f64(c, c->out_stage.as<double>(), ang, (size_t)tot[1], dtype, mem, 1));
    if (mem == WOL_MEM_HOST) HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(hipGetLastError());
    return WOL_OK;
}

int wol_cos_angs(wol_ctx* c, const double* pos, int n, const double* sub, int ns, const double* box,
                 double lowcut, double highcut, double* ang, double* num_angs, int64_t* total) {
    return wol_cos_angs_t(c, pos, n, sub, ns, box, lowcut, highcut, ang, num_angs, total, WOL_F64, WOL_MEM_HOST);
}

int wol_lsi_frame_t(wol_ctx* c, const void* pos, int n, const void* sub, int ns, const double* box,
                    double lowcut, double highcut, void* lsi_full, void* num_lsi, int64_t* nvalid, int dtype, int mem) {
    WOL_TRY(check_ctx(c));
    WOL_TRY(check_real(dtype, mem));
    if (!pos || !box || !lsi_full || n < 0 || ns < 0) return fail(WOL_ERR_INVALID, "lsi_frame: bad argument");
    if (!sub) ns = n;
    if (nvalid) *nvalid = 0;
    if (ns == 0) return WOL_OK;
    if (n == 0) {
        if (mem != WOL_MEM_HOST) return fail(WOL_ERR_INVALID, "lsi_frame: n == 0 with device buffers is not supported");
        for (int i = 0; i < ns; ++i) {
            if (dtype == WOL_F64) { static_cast<double*>(lsi_full)[i] = NAN; if (num_lsi) static_cast<double*>(num_lsi)[i] = 0.0; }
            else { static_cast<float*>(lsi_full)[i] = NAN; if (num_lsi) static_cast<float*>(num_lsi)[i] = 0.f; }
        }
        return WOL_OK;
    }
    const double high2 = highcut + 3.7;  // water_properties.py:276,281
    ShellArgs sa;
    WOL_TRY(shell_setup(c, RealSrc{pos, dtype, mem}, n, RealSrc{sub, dtype, mem}, ns, box, high2, &sa));
    sa.lowsq = lowcut * lowcut;
    sa.highsq = highcut * highcut;
    WOL_TRY(c->out_stage.ensure(sizeof(double) * 2 * (size_t)ns));
    WOL_TRY(c->slow_count.ensure(sizeof(int) * 4));
    double* d_lsi = c->out_stage.as<double>();
    double* d_num = d_lsi + ns;
    int over = 0;
    // the shell's distances are sorted in a per-water scratch row: a first attempt with room for a liquid's shell many
    // times over; a water with more neighbours than that (a dense cluster, a long highcut) reports how many it has and
    // the kernel runs once more with rows of that length
    for (int cap = WOL_LSI_CAP, attempt = 0; attempt < 2; ++attempt) {
        WOL_TRY(c->qbuf.ensure(sizeof(double) * (size_t)cap * (size_t)ns));
        HIP_TRY(hipMemsetAsync(c->slow_count.p, 0, sizeof(int) * 4, c->stream));
        hipLaunchKernelGGL(k_lsi, dim3((ns + WOL_BLOCK - 1) / WOL_BLOCK), dim3(WOL_BLOCK), 0, c->stream, sa,
                           high2 * high2, c->qbuf.as<double>(), cap, d_lsi, d_num, c->slow_count.as<int>());
        HIP_TRY(hipMemcpyAsync(&over, c->slow_count.p, sizeof(int), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        if (!over) break;
        if ((size_t)over * (size_t)ns * sizeof(double) > ((size_t)16 << 30))
            return fail(WOL_ERR_NOMEM, "lsi_frame: a water has %d neighbours inside highcut; %d such rows do not fit", over - 2, ns);
        cap = over;
    }
    std::vector<double> hlsi((size_t)ns);   // the count of valid entries (len(lsiVals), wp:308) is a host value
    WOL_TRY(deliver_f64(c, d_lsi, lsi_full, (size_t)ns, dtype, mem, 0));
    WOL_TRY(deliver_f64(c, d_num, num_lsi, (size_t)ns, dtype, mem, 1));
    HIP_TRY(hipMemcpyAsync(hlsi.data(), d_lsi, sizeof(double) * (size_t)ns, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(hipGetLastError());
    if (over) return fail(WOL_ERR_INVALID, "lsi_frame: shell larger than reported (%d)", over);
    int64_t nv = 0;
    for (int i = 0; i < ns; ++i) nv += (hlsi[(size_t)i] == hlsi[(size_t)i]) ? 1 : 0;
    if (nvalid) *nvalid = nv;
    return WOL_OK;
}

int wol_lsi_frame(wol_ctx* c, const double* pos, int n, const double* sub, int ns, const double* box,
                  double lowcut, double highcut, double* lsi_full, double* num_lsi, int64_t* nvalid) {
    return wol_lsi_frame_t(c, pos, n, sub, ns, box, lowcut, highcut, lsi_full, num_lsi, nvalid, WOL_F64, WOL_MEM_HOST);
}

// ---- Willard-Chandler density field ---------------------------------------
static int willard_impl(wol_ctx* c, const double* pos, int n, const double* pts, int64_t npts, const double* gx,
                        int nx, const double* gy, int ny, const double* gz, int nz, const double* box,
                        double smoothlen, double* densvals, double* densnorms) {
    WOL_TRY(check_ctx(c));
    if (!pos || !box || !densvals || !densnorms || n < 0 || npts < 0) return fail(WOL_ERR_INVALID, "willard: bad argument");
    if (!(smoothlen > 0.0)) return fail(WOL_ERR_INVALID, "willard: smoothlen must be > 0");
    if (npts == 0) return WOL_OK;
    if (npts > 0x7fffffffLL) return fail(WOL_ERR_INVALID, "willard: too many points");
    if (n == 0) {  // no waters: zero density, 0/0 normals (what the reference's empty loop leaves)
        for (int64_t i = 0; i < npts; ++i) densvals[i] = 0.0;
        for (int64_t i = 0; i < 3 * npts; ++i) densnorms[i] = std::nan("");
        return WOL_OK;
    }
    ShellArgs sa;
    // the `sub` slot of the set-up carries the query points in points mode
    WOL_TRY(shell_setup(c, pos, n, pts, pts ? (int)npts : 0, box, 3.0 * smoothlen, &sa));
    sa.lowsq = 0.0; sa.highsq = 0.0;
    WillardArgs w;
    w.gx = w.gy = w.gz = nullptr;
    w.nx = nx; w.ny = ny; w.nz = nz; w.smoothlen = smoothlen;
    const size_t ob = sizeof(double) * 4 * (size_t)npts;
    WOL_TRY(c->out_stage.ensure(ob));
    w.densvals = c->out_stage.as<double>();
    w.densnorms = w.densvals + npts;
    if (!pts) {
        WOL_TRY(c->small.ensure(sizeof(double) * (size_t)(nx + ny + nz)));
        double* d = c->small.as<double>();
        HIP_TRY(hipMemcpyAsync(d, gx, sizeof(double) * nx, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipMemcpyAsync(d + nx, gy, sizeof(double) * ny, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipMemcpyAsync(d + nx + ny, gz, sizeof(double) * nz, hipMemcpyHostToDevice, c->stream));
        w.gx = d; w.gy = d + nx; w.gz = d + nx + ny;
    }
    std::pair<hipEvent_t, hipEvent_t> ev;
    if (c->time_kernels) WOL_TRY(begin_timing(c, &ev));
    hipLaunchKernelGGL(k_willard, dim3((unsigned)((npts + WOL_BLOCK - 1) / WOL_BLOCK)), dim3(WOL_BLOCK), 0, c->stream, sa, w);
    if (c->time_kernels) WOL_TRY(end_timing(c, ev));
    HIP_TRY(hipMemcpyAsync(densvals, w.densvals, sizeof(double) * (size_t)npts, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipMemcpyAsync(densnorms, w.densnorms, sizeof(double) * 3 * (size_t)npts, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(hipGetLastError());
    return WOL_OK;
}

int wol_willard_density_field(wol_ctx* c, const double* pos, int n, const double* gridx, int nx, const double* gridy,
                              int ny, const double* gridz, int nz, const double* box, double smoothlen,
                              double* densvals, double* densnorms) {
    if (!gridx || !gridy || !gridz || nx < 0 || ny < 0 || nz < 0) return fail(WOL_ERR_INVALID, "willard: bad grid");
    return willard_impl(c, pos, n, nullptr, (int64_t)nx * ny * nz, gridx, nx, gridy, ny, gridz, nz, box, smoothlen,
                        densvals, densnorms);
}

int wol_willard_density_points(wol_ctx* c, const double* pos, int n, const double* pts, int npts, const double* box,
                               double smoothlen, double* densvals, double* densnorms) {
    if (!pts && npts > 0) return fail(WOL_ERR_INVALID, "willard: pts must not be NULL");
    return willard_impl(c, pos, n, pts, npts, nullptr, 0, nullptr, 0, nullptr, 0, box, smoothlen, densvals, densnorms);
}

// ---- water orientation ---------------------------------------------------------
static int orient_impl(wol_ctx* c, const double* opos, const double* hpos, int n, const double* refvec, const double* box,
                       double* out0, double* out1) {
    WOL_TRY(check_ctx(c));
    if (!opos || !hpos || !box || !out0 || n < 0) return fail(WOL_ERR_INVALID, "orientation: bad argument");
    if (n == 0) return WOL_OK;
    const size_t nb = sizeof(double) * 3 * (size_t)n;
    WOL_TRY(c->in_stage.ensure(3 * nb));
    WOL_TRY(c->out_stage.ensure(nb));
    double* d_o = c->in_stage.as<double>();
    double* d_h = d_o + 3 * (size_t)n;
    double* d_out = c->out_stage.as<double>();
    HIP_TRY(hipMemcpyAsync(d_o, opos, nb, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(d_h, hpos, 2 * nb, hipMemcpyHostToDevice, c->stream));
    const dim3 grid((n + 255) / 256);
    if (refvec) {
        hipLaunchKernelGGL(k_wat_orient, grid, dim3(256), 0, c->stream, d_o, d_h, n, refvec[0], refvec[1], refvec[2],
                           make_box(box), d_out, d_out + n);
        HIP_TRY(hipMemcpyAsync(out0, d_out, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipMemcpyAsync(out1, d_out + n, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost, c->stream));
    } else {
        hipLaunchKernelGGL(k_wat_oh_vec, grid, dim3(256), 0, c->stream, d_o, d_h, n, make_box(box), d_out);
        HIP_TRY(hipMemcpyAsync(out0, d_out, nb, hipMemcpyDeviceToHost, c->stream));
    }
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(hipGetLastError());
    return WOL_OK;
}

int wol_wat_orient(wol_ctx* c, const double* opos, const double* hpos, int n, const double* refvec, const double* box,
                   double* angdip, double* angplane) {
    if (!refvec || !angplane) return fail(WOL_ERR_INVALID, "watorient: bad argument");
    if (!((refvec[0] * refvec[0] + refvec[1] * refvec[1]) + refvec[2] * refvec[2] > 0.0))
        return fail(WOL_ERR_INVALID, "watorient: reference vector must not be zero");
    return orient_impl(c, opos, hpos, n, refvec, box, angdip, angplane);
}

int wol_wat_oh_vec(wol_ctx* c, const double* opos, const double* hpos, int n, const double* box, double* normdip) {
    return orient_impl(c, opos, hpos, n, nullptr, box, normdip, nullptr);
}

// ---- radial distribution functions --------------------------------------------
// counts of pair distances per bin: candidates = pos2 (cell-sorted), queries = pos1 (null: pos2 itself, pairs once)
static int pair_counts_impl(wol_ctx* c, const double* pos1, int n1, const double* pos2, int n2, double binwidth,
                            int totbins, const double* box, std::vector<long long>* counts) {
    WOL_TRY(check_ctx(c));
    if (!pos2 || !box || n1 < 0 || n2 < 0) return fail(WOL_ERR_INVALID, "rdf: bad argument");
    if (!(binwidth > 0.0) || totbins <= 0 || totbins > 12000) return fail(WOL_ERR_INVALID, "rdf: need binwidth > 0 and 0 < totbins <= 12000");
    counts->assign((size_t)totbins, 0);
    const int nq = pos1 ? n1 : n2;
    if (n2 == 0 || nq == 0) return WOL_OK;
    ShellArgs sa;
    WOL_TRY(shell_setup(c, pos2, n2, pos1, n1, box, binwidth * (double)totbins, &sa));
    WOL_TRY(c->hist_stage.ensure(sizeof(unsigned long long) * (size_t)totbins));
    HIP_TRY(hipMemsetAsync(c->hist_stage.p, 0, sizeof(unsigned long long) * (size_t)totbins, c->stream));
    const size_t lds = sizeof(int) * (size_t)totbins;
    int grid = 0;
    std::pair<hipEvent_t, hipEvent_t> ev;
    const FrameGrid* gh = c->grids_host.as<FrameGrid>();
    const double reach = binwidth * (double)totbins;
    if (!pos1 && sa.use_cells && c->pair_tiled && lds <= 16 * 1024 && reach < 60.0) {
        // one set, pairs once: the LDS-tiled kernel (runs of cells along x, about a hundred waters per 128-lane block)
        PairTileArgs pa;
        pa.grid = sa.grid; pa.sorted = sa.sorted; pa.cell_start = sa.cell_start; pa.n = n2;
        const double wpc = std::fmax((double)n2 / (double)gh->ncell, 0.05);
        int cxrun = (int)std::fmax(1.0, std::floor(58.0 / wpc));
        cxrun = std::min(cxrun, gh->nc[0]);
        pa.ntx = (gh->nc[0] + cxrun - 1) / cxrun;
        pa.cxrun = (gh->nc[0] + pa.ntx - 1) / pa.ntx;      // equalised
        // fp32 coordinates are taken relative to the run's centre (nearest image to the centre): that is every lane's
        // nearest image when no axis is shorter than twice (the tile's half extent + the reach)
        pa.need_minimg = 0;
        const double ext[3] = {0.5 * pa.cxrun * gh->cw[0], 0.5 * gh->cw[1], 0.5 * gh->cw[2]};
        for (int k = 0; k < 3; ++k)
            if (gh->box.L[k] < 2.0 * (ext[k] + reach) + 1.0) pa.need_minimg = 1;
        pa.binwidth = binwidth; pa.totbins = totbins;
        pa.cut2f = (float)(reach * reach * (1.0 + 1e-5) + 1e-3);
        pa.hist = c->hist_stage.as<unsigned long long>();
        const int64_t units = (int64_t)gh->nc[1] * gh->nc[2] * pa.ntx * (gh->lo[1] + gh->hi[1] + 1) * (gh->lo[2] + gh->hi[2] + 1);
        if (pa.need_minimg) WOL_TRY(resident_grid(c, k_pair_hist_tiled<true>, WOL_PT_BLOCK, lds, units, &grid));
        else WOL_TRY(resident_grid(c, k_pair_hist_tiled<false>, WOL_PT_BLOCK, lds, units, &grid));
        if (c->time_kernels) WOL_TRY(begin_timing(c, &ev));
        if (pa.need_minimg) hipLaunchKernelGGL(k_pair_hist_tiled<true>, dim3(grid), dim3(WOL_PT_BLOCK), lds, c->stream, pa);
        else hipLaunchKernelGGL(k_pair_hist_tiled<false>, dim3(grid), dim3(WOL_PT_BLOCK), lds, c->stream, pa);
        if (c->time_kernels) WOL_TRY(end_timing(c, ev));
    } else {
        WOL_TRY(resident_grid(c, k_pair_hist, WOL_BLOCK, lds, (nq + WOL_BLOCK - 1) / WOL_BLOCK, &grid));
        if (c->time_kernels) WOL_TRY(begin_timing(c, &ev));
        hipLaunchKernelGGL(k_pair_hist, dim3(grid), dim3(WOL_BLOCK), lds, c->stream, sa, pos1 ? 0 : 1, binwidth, totbins,
                           c->hist_stage.as<unsigned long long>());
        if (c->time_kernels) WOL_TRY(end_timing(c, ev));
    }
    HIP_TRY(hipMemcpyAsync(counts->data(), c->hist_stage.p, sizeof(long long) * (size_t)totbins, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(hipGetLastError());
    return WOL_OK;
}

int wol_radial_dist_same(wol_ctx* c, const double* pos, int n, double binwidth, int totbins, double bulkdens,
                         const double* box, double* rdf) {
    if (!rdf) return fail(WOL_ERR_INVALID, "rdf: output must not be NULL");
    if (totbins > 1290) return fail(WOL_ERR_INVALID, "rdf: totbins > 1290 overflows the reference's integer k**3");
    std::vector<long long> counts;
    WOL_TRY(pair_counts_impl(c, nullptr, 0, pos, n, binwidth, totbins, box, &counts));
    rdf_normalise(counts, n, bulkdens, binwidth, rdf);
    return WOL_OK;
}

int wol_radial_dist(wol_ctx* c, const double* pos1, int n1, const double* pos2, int n2, double binwidth, int totbins,
                    double bulkdens, const double* box, double* rdf) {
    if (!rdf || !pos1) return fail(WOL_ERR_INVALID, "rdf: pos1 / output must not be NULL");
    if (totbins > 1290) return fail(WOL_ERR_INVALID, "rdf: totbins > 1290 overflows the reference's integer k**3");
    std::vector<long long> counts;
    WOL_TRY(pair_counts_impl(c, pos1, n1, pos2, n2, binwidth, totbins, box, &counts));
    rdf_normalise(counts, n1, bulkdens, binwidth, rdf);
    return WOL_OK;
}

int wol_pair_distance_histogram(wol_ctx* c, const double* pos1, int n1, const double* pos2, int n2, double binwidth,
                                int totbins, const double* box, double* hist) {
    if (!hist || !pos1) return fail(WOL_ERR_INVALID, "pair histogram: pos1 / output must not be NULL");
    std::vector<long long> counts;
    WOL_TRY(pair_counts_impl(c, pos1, n1, pos2, n2, binwidth, totbins, box, &counts));
    for (int k = 0; k < totbins; ++k) hist[k] = (double)counts[(size_t)k];
    return WOL_OK;
}

// ---- hydrogen bonds --------------------------------------------------------------
// the cut-off angle with cos / sin of its radian value as double-doubles (~106 bits), for the device's exact decision
// next to the cut (wol_ang_below_cut).  Minimal double-double arithmetic (error-free sums and fma products) and the
// Taylor series of sin / cos about pi/2.
namespace ddm {
struct dd { double hi, lo; };
inline dd two_sum(double a, double b) { const double s = a + b, bb = s - a; return {s, (a - (s - bb)) + (b - bb)}; }
inline dd quick(double a, double b) { const double s = a + b; return {s, b - (s - a)}; }
inline dd add(dd a, dd b) { dd s = two_sum(a.hi, b.hi); s.lo += a.lo + b.lo; return quick(s.hi, s.lo); }
inline dd neg(dd a) { return {-a.hi, -a.lo}; }
inline dd mul(dd a, dd b) {
    const double p = a.hi * b.hi, e = std::fma(a.hi, b.hi, -p) + (a.hi * b.lo + a.lo * b.hi);
    return quick(p, e);
}
inline dd div_d(dd a, double b) {   // a / b, b a small integer
    const double q1 = a.hi / b;
    const double r = std::fma(-q1, b, a.hi) + a.lo;
    return quick(q1, r / b);
}
// sin and cos of x, |x| <= pi/2, Taylor series to x^41 (remainder below 1e-33)
inline void sin_cos(dd x, dd* sn, dd* cs) {
    const dd x2 = mul(x, x);
    dd ts = x, tc = {1.0, 0.0}, s = x, c = {1.0, 0.0};
    for (int k = 1; k <= 20; ++k) {
        tc = neg(div_d(mul(tc, x2), (double)((2 * k - 1) * (2 * k))));
        ts = neg(div_d(mul(ts, x2), (double)((2 * k) * (2 * k + 1))));
        c = add(c, tc);
        s = add(s, ts);
    }
    *sn = s; *cs = c;
}
}  // namespace ddm

static AngCut make_ang_cut(double angcut) {
    AngCut k;
    k.cut_deg = angcut;
    k.A = angcut * (WOL_PI / 180.0);
    k.refine = (angcut >= 1.0 && angcut <= 179.0) ? 1 : 0;
    // A = pi/2 + x: cos A = -sin x, sin A = cos x
    const ddm::dd half_pi = {1.5707963267948966, 6.123233995736766e-17};
    const ddm::dd x = ddm::add(ddm::dd{k.A, 0.0}, ddm::neg(half_pi));
    ddm::dd sn, cs;
    ddm::sin_cos(x, &sn, &cs);
    k.cA_hi = -sn.hi; k.cA_lo = -sn.lo;
    k.sA = cs.hi;
    return k;
}

int wol_general_hbonds(wol_ctx* c, const double* acc, int nacc, const double* don, const double* donh, int ndon,
                       const double* box, double distcut, double angcut, int32_t* bond) {
    WOL_TRY(check_ctx(c));
    if (!acc || !don || !donh || !box || !bond || nacc < 0 || ndon < 0) return fail(WOL_ERR_INVALID, "generalhbonds: bad argument");
    const int64_t total = (int64_t)nacc * ndon;
    if (total == 0) return WOL_OK;
    if (total > (int64_t)1 << 33) return fail(WOL_ERR_INVALID, "generalhbonds: a %d x %d mask is too large; use wol_general_hbonds_list", nacc, ndon);
    const size_t ab = sizeof(double) * 3 * (size_t)nacc, db = sizeof(double) * 3 * (size_t)ndon;
    WOL_TRY(c->in_stage.ensure(ab + 2 * db));
    WOL_TRY(c->out_stage.ensure(sizeof(int32_t) * (size_t)total));
    double* d_a = c->in_stage.as<double>();
    double* d_d = d_a + 3 * (size_t)nacc;
    double* d_h = d_d + 3 * (size_t)ndon;
    HIP_TRY(hipMemcpyAsync(d_a, acc, ab, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(d_d, don, db, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(d_h, donh, db, hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(k_hbonds_dense, dim3((unsigned)((total + WOL_BLOCK - 1) / WOL_BLOCK)), dim3(WOL_BLOCK), 0, c->stream,
                       d_a, nacc, d_d, d_h, ndon, make_box(box), distcut * distcut, make_ang_cut(angcut), c->out_stage.as<int32_t>());
    HIP_TRY(hipMemcpyAsync(bond, c->out_stage.p, sizeof(int32_t) * (size_t)total, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(hipGetLastError());
    return WOL_OK;
}

int wol_general_hbonds_list(wol_ctx* c, const double* acc, int nacc, const double* don, const double* donh, int ndon,
                            const double* box, double distcut, double angcut, int64_t capacity, int32_t* acc_idx,
                            int32_t* don_idx, int64_t* count) {
    WOL_TRY(check_ctx(c));
    if (!acc || !don || !donh || !box || !count || nacc < 0 || ndon < 0 || capacity < 0 || (capacity > 0 && (!acc_idx || !don_idx)))
        return fail(WOL_ERR_INVALID, "generalhbonds list: bad argument");
    if (!(distcut > 0.0)) return fail(WOL_ERR_INVALID, "generalhbonds list: distCut must be > 0");
    *count = 0;
    if (nacc == 0 || ndon == 0) return WOL_OK;
    ShellArgs sa;
    WOL_TRY(shell_setup(c, don, ndon, acc, nacc, box, distcut, &sa));
    WOL_TRY(c->out_stage.ensure(sizeof(double) * 3 * (size_t)ndon));
    WOL_TRY(c->slow_list.ensure(sizeof(unsigned long long) * (size_t)std::max<int64_t>(capacity, 1)));
    WOL_TRY(c->small.ensure(sizeof(unsigned long long)));
    HIP_TRY(hipMemcpyAsync(c->out_stage.p, donh, sizeof(double) * 3 * (size_t)ndon, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemsetAsync(c->small.p, 0, sizeof(unsigned long long), c->stream));
    hipLaunchKernelGGL(k_hbonds_list, dim3((nacc + WOL_BLOCK - 1) / WOL_BLOCK), dim3(WOL_BLOCK), 0, c->stream, sa,
                       c->out_stage.as<double>(), distcut * distcut, make_ang_cut(angcut), c->slow_list.as<unsigned long long>(),
                       (unsigned long long)capacity, c->small.as<unsigned long long>());
    unsigned long long found = 0;
    HIP_TRY(hipMemcpyAsync(&found, c->small.p, sizeof found, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(hipGetLastError());
    *count = (int64_t)found;   // may exceed capacity: call again with a larger buffer
    const size_t got = (size_t)std::min<unsigned long long>(found, (unsigned long long)capacity);
    if (got == 0) return WOL_OK;
    std::vector<unsigned long long> pairs(got);
    HIP_TRY(hipMemcpy(pairs.data(), c->slow_list.p, sizeof(unsigned long long) * got, hipMemcpyDeviceToHost));
    std::sort(pairs.begin(), pairs.end());   // (acceptor, donor) ascending: the order of np.where(bondBool)
    for (size_t k = 0; k < got; ++k) { acc_idx[k] = (int32_t)(pairs[k] >> 32); don_idx[k] = (int32_t)(pairs[k] & 0xFFFFFFFFu); }
    return WOL_OK;
}

// ---- level 3 ------------------------------------------------------------------------------
int wol_q_traj(wol_ctx* c, const double* frames, int64_t atom_stride, int64_t frame_stride, int T, int n,
               const double* boxes, double lowcut, double highcut, double* mean, double* var, int64_t* qhist,
               int qbins, int64_t* dhist, int dbins, double dbinwidth, double* q_out, int mem) {
    return q_traj_impl<double>(c, frames, atom_stride, frame_stride, T, n, boxes, lowcut, highcut, mean, var,
                               qhist, qbins, dhist, dbins, dbinwidth, q_out, mem);
}

int wol_q_traj_f32(wol_ctx* c, const float* frames, int64_t atom_stride, int64_t frame_stride, int T, int n,
                   const double* boxes, double lowcut, double highcut, double* mean, double* var, int64_t* qhist,
                   int qbins, int64_t* dhist, int dbins, double dbinwidth, float* q_out, int mem) {
    return q_traj_impl<float>(c, frames, atom_stride, frame_stride, T, n, boxes, lowcut, highcut, mean, var,
                              qhist, qbins, dhist, dbins, dbinwidth, q_out, mem);
}

int wol_q_traj_pops(wol_ctx* c, const void* frames, int64_t atom_stride, int64_t frame_stride, int T, int n, const double* boxes,
                    double lowcut, double highcut, double* mean, double* var, int64_t* qhist, int qbins, int64_t* dhist, int dbins,
                    double dbinwidth, void* q_out, const void* pop_pos, const int64_t* pop_off, int npops, double* pop_mean,
                    double* pop_var, int64_t* pop_qhist, void* pop_q, int dtype, int mem) {
    WOL_TRY(check_real(dtype, mem));
    PopQuery pq;
    pq.pos = pop_pos; pq.off = pop_off; pq.npops = npops;
    if (dtype == WOL_F64)
        return q_traj_impl<double>(c, static_cast<const double*>(frames), atom_stride, frame_stride, T, n, boxes, lowcut, highcut, mean, var,
                                   qhist, qbins, dhist, dbins, dbinwidth, static_cast<double*>(q_out), mem, &pq, pop_mean, pop_var, pop_qhist,
                                   static_cast<double*>(pop_q));
    return q_traj_impl<float>(c, static_cast<const float*>(frames), atom_stride, frame_stride, T, n, boxes, lowcut, highcut, mean, var,
                              qhist, qbins, dhist, dbins, dbinwidth, static_cast<float*>(q_out), mem, &pq, pop_mean, pop_var, pop_qhist,
                              static_cast<float*>(pop_q));
}

int wol_cos_angs_traj(wol_ctx* c, const void* frames, int64_t atom_stride, int64_t frame_stride, int T, int n, const double* boxes,
                      double lowcut, double highcut, const void* pop_pos, const int64_t* pop_off, int npops, int nbins,
                      int64_t* ang_hist, int64_t* counts, double* cos_sums, int dtype, int mem) {
    WOL_TRY(check_real(dtype, mem));
    PopQuery pq;
    pq.pos = pop_pos; pq.off = pop_off; pq.npops = npops;
    ShellTrajOut o;
    o.kind = 0; o.nbins = nbins; o.hist = ang_hist; o.counts = counts; o.a = cos_sums;
    if (dtype == WOL_F64)
        return shell_traj_impl<double>(c, static_cast<const double*>(frames), atom_stride, frame_stride, T, n, boxes, lowcut, highcut, pq, o, mem);
    return shell_traj_impl<float>(c, static_cast<const float*>(frames), atom_stride, frame_stride, T, n, boxes, lowcut, highcut, pq, o, mem);
}

int wol_radial_dist_same_traj(wol_ctx* c, const void* frames, int64_t atom_stride, int64_t frame_stride, int T, int n, const double* boxes,
                              double binwidth, int totbins, double bulkdens, double* rdf, int64_t* counts, int dtype, int mem) {
    WOL_TRY(check_real(dtype, mem));
    if (dtype == WOL_F64)
        return pair_hist_traj_impl<double>(c, static_cast<const double*>(frames), atom_stride, frame_stride, T, n, boxes, binwidth, totbins,
                                           bulkdens, rdf, counts, mem);
    return pair_hist_traj_impl<float>(c, static_cast<const float*>(frames), atom_stride, frame_stride, T, n, boxes, binwidth, totbins, bulkdens,
                                      rdf, counts, mem);
}

int wol_lsi_traj(wol_ctx* c, const void* frames, int64_t atom_stride, int64_t frame_stride, int T, int n, const double* boxes,
                 double lowcut, double highcut, const void* pop_pos, const int64_t* pop_off, int npops, int nbins, double hist_max,
                 int64_t* lsi_hist, int64_t* counts, double* mean, double* var, void* lsi_out, void* pop_lsi, int dtype, int mem) {
    WOL_TRY(check_real(dtype, mem));
    PopQuery pq;
    pq.pos = pop_pos; pq.off = pop_off; pq.npops = npops;
    ShellTrajOut o;
    o.kind = 1; o.nbins = nbins; o.hist_max = hist_max; o.hist = lsi_hist; o.counts = counts; o.a = mean; o.b = var;
    o.per_water = lsi_out; o.per_point = pop_lsi;
    if (dtype == WOL_F64)
        return shell_traj_impl<double>(c, static_cast<const double*>(frames), atom_stride, frame_stride, T, n, boxes, lowcut, highcut, pq, o, mem);
    return shell_traj_impl<float>(c, static_cast<const float*>(frames), atom_stride, frame_stride, T, n, boxes, lowcut, highcut, pq, o, mem);
}

}  // extern "C"

// ---- multi-GPU: frame shards + the path's one exchange step -----------------------------
// RCCL is bound at run time: whichever librccl.so.1 the process already maps (PyTorch-ROCm ships its own copy, which
// must be the one used next to torch) or, failing that, the system's.
struct wol_comm {
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1;
    bool own = false;
};

namespace {
struct RcclApi {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};
RcclApi g_rccl;

int rccl_load() {
    if (g_rccl.handle) return WOL_OK;
    void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_NOLOAD);
    if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) return fail(WOL_ERR_HIP, "RCCL not found: %s", dlerror());
    RcclApi a;
    a.handle = h;
    a.GetUniqueId = reinterpret_cast<decltype(a.GetUniqueId)>(dlsym(h, "ncclGetUniqueId"));
    a.CommInitRank = reinterpret_cast<decltype(a.CommInitRank)>(dlsym(h, "ncclCommInitRank"));
    a.CommDestroy = reinterpret_cast<decltype(a.CommDestroy)>(dlsym(h, "ncclCommDestroy"));
    a.AllReduce = reinterpret_cast<decltype(a.AllReduce)>(dlsym(h, "ncclAllReduce"));
    a.AllGather = reinterpret_cast<decltype(a.AllGather)>(dlsym(h, "ncclAllGather"));
    a.GetErrorString = reinterpret_cast<decltype(a.GetErrorString)>(dlsym(h, "ncclGetErrorString"));
    if (!a.GetUniqueId || !a.CommInitRank || !a.CommDestroy || !a.AllReduce || !a.AllGather)
        return fail(WOL_ERR_HIP, "RCCL library lacks an entry point");
    g_rccl = a;
    return WOL_OK;
}

#define NCCL_TRY(expr)                                                                                   \
    do {                                                                                                 \
        ncclResult_t r_ = (expr);                                                                        \
        if (r_ != ncclSuccess)                                                                           \
            return fail(WOL_ERR_HIP, "%s: %s", #expr, g_rccl.GetErrorString ? g_rccl.GetErrorString(r_) : "RCCL error"); \
    } while (0)

template <typename T>
int q_traj_sharded_impl(wol_ctx* c, wol_comm* comm, const T* frames, int64_t atom_stride, int64_t frame_stride, int T_local,
                        int T_total, int n, const double* boxes, double low, double high, double* mean, double* var,
                        int64_t* qhist, int qbins, int64_t* dhist, int dbins, double dbinwidth, T* q_out, int mem) {
    // Failures that keep this rank out of the exchange altogether (no device, no communicator, no exchange buffer) return at
    // once -- the caller must then abort the communicator, as with any collective.  EVERY other failure of this rank (bad
    // arguments, a failed copy, a failed kernel launch, out of memory in the pipeline) is held back until after the exchange:
    // the rank contributes zeros plus a status word that rides on the histogram all-reduce, so that all ranks leave the
    // collectives together and all of them return an error (this rank its own, the others WOL_ERR_PEER).
    WOL_TRY(check_ctx(c));
    if (!comm) return fail(WOL_ERR_INVALID, "q_traj_sharded: communicator must not be NULL");
    const bool host = (mem == WOL_MEM_HOST);
    const int W = comm->world;
    const int Tt = std::max(T_total, 0);
    const int per = (Tt + W - 1) / W;                 // frames per rank, padded: shards differ by at most one frame
    const int nq = (qhist && qbins > 0) ? qbins : 0, nd = (dhist && dbins > 0) ? dbins : 0;
    const int nh = nq + nd;
    // exchange buffer (device): hist[nh] uint64 | status[1] uint64 | mine[per][2] doubles | all[W][per][2] doubles
    const size_t o_mine = sizeof(unsigned long long) * (size_t)(nh + 1);
    const size_t o_all = o_mine + sizeof(double) * 2 * (size_t)per;
    WOL_TRY(c->xchg.ensure(o_all + sizeof(double) * 2 * (size_t)per * W));
    char* xb = c->xchg.as<char>();
    unsigned long long* d_hist = reinterpret_cast<unsigned long long*>(xb);
    unsigned long long* d_status = d_hist + nh;
    double* d_mine = reinterpret_cast<double*>(xb + o_mine);
    double* d_all = reinterpret_cast<double*>(xb + o_all);
    HIP_TRY(hipMemsetAsync(xb, 0, o_all, c->stream));
    // ---- this rank's frames: the local pipeline, integer histograms and the per-frame series left on the device ----
    auto local = [&]() -> int {
        if (mem != WOL_MEM_HOST && mem != WOL_MEM_DEVICE) return fail(WOL_ERR_INVALID, "bad mem space");
        int t0 = 0, t1 = 0;
        WOL_TRY(wol_shard_bounds(T_total, comm->rank, comm->world, &t0, &t1));
        if (T_local != t1 - t0)
            return fail(WOL_ERR_INVALID, "q_traj_sharded: rank %d of %d owns frames [%d, %d) of %d, got %d frames", comm->rank,
                        comm->world, t0, t1, T_total, T_local);
        if (qhist && qbins <= 0) return fail(WOL_ERR_INVALID, "qbins must be > 0");
        if (dhist && (dbins <= 0 || !(dbinwidth > 0.0))) return fail(WOL_ERR_INVALID, "dbins/dbinwidth must be > 0");
        if (T_local <= 0) return WOL_OK;
        if (host) {
            std::vector<int64_t> hq((size_t)std::max(qbins, 1), 0), hd((size_t)std::max(dbins, 1), 0);
            std::vector<double> hm((size_t)T_local), hv((size_t)T_local);
            WOL_TRY((q_traj_impl<T>(c, frames, atom_stride, frame_stride, T_local, n, boxes, low, high, hm.data(), hv.data(),
                                    qhist ? hq.data() : nullptr, qbins, dhist ? hd.data() : nullptr, dbins, dbinwidth, q_out, mem)));
            std::vector<double> mv(2 * (size_t)per, 0.0);
            for (int t = 0; t < T_local; ++t) { mv[2 * (size_t)t] = hm[(size_t)t]; mv[2 * (size_t)t + 1] = hv[(size_t)t]; }
            if (qhist) HIP_TRY(hipMemcpyAsync(d_hist, hq.data(), sizeof(int64_t) * (size_t)qbins, hipMemcpyHostToDevice, c->stream));
            if (dhist) HIP_TRY(hipMemcpyAsync(d_hist + nq, hd.data(), sizeof(int64_t) * (size_t)dbins, hipMemcpyHostToDevice, c->stream));
            HIP_TRY(hipMemcpyAsync(d_mine, mv.data(), sizeof(double) * mv.size(), hipMemcpyHostToDevice, c->stream));
            HIP_TRY(hipStreamSynchronize(c->stream));   // the host vectors above go out of scope
        } else {
            // mean / var land in a scratch pair of arrays, then interleave into (mean, var) rows
            WOL_TRY(c->cvt_out[3].ensure(sizeof(double) * 2 * (size_t)T_local));
            double* sm = c->cvt_out[3].as<double>();
            WOL_TRY((q_traj_impl<T>(c, frames, atom_stride, frame_stride, T_local, n, boxes, low, high, sm, sm + T_local,
                                    qhist ? reinterpret_cast<int64_t*>(d_hist) : nullptr, qbins,
                                    dhist ? reinterpret_cast<int64_t*>(d_hist + nq) : nullptr, dbins, dbinwidth,
                                    q_out, mem)));
            HIP_TRY(hipMemcpy2DAsync(d_mine, 2 * sizeof(double), sm, sizeof(double), sizeof(double), (size_t)T_local, hipMemcpyDeviceToDevice, c->stream));
            HIP_TRY(hipMemcpy2DAsync(d_mine + 1, 2 * sizeof(double), sm + T_local, sizeof(double), sizeof(double), (size_t)T_local, hipMemcpyDeviceToDevice, c->stream));
        }
        return WOL_OK;
    };
    const int local_rc = local();
    const std::string local_msg = g_err;
    if (local_rc != WOL_OK) {
        // nothing half-done goes into the sums: zeros and a raised status word instead
        (void)hipGetLastError();
        HIP_TRY(hipMemsetAsync(xb, 0, o_all, c->stream));
        const unsigned long long one = 1ull;
        HIP_TRY(hipMemcpyAsync(d_status, &one, sizeof one, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
    }
    // ---- the exchange step: integer sums (order-independent, so the result is the one-GPU result bit for bit; the last
    //      word counts the ranks whose local part failed) and the gathered per-frame series; KB-sized, latency-bound ----
    NCCL_TRY(g_rccl.AllReduce(d_hist, d_hist, (size_t)nh + 1, ncclUint64, ncclSum, comm->comm, c->stream));
    if (mean || var) NCCL_TRY(g_rccl.AllGather(d_mine, d_all, 2 * (size_t)per, ncclFloat64, comm->comm, c->stream));
    // the agreed status is a host decision (one word; the call synchronises here in either memory mode)
    unsigned long long failed = 0;
    HIP_TRY(hipMemcpyAsync(&failed, d_status, sizeof failed, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (local_rc != WOL_OK) { g_err = local_msg; return local_rc; }
    if (failed != 0)
        return fail(WOL_ERR_PEER, "q_traj_sharded: %llu of %d ranks failed in their local part; no result was written", failed, W);
    // ---- results to the caller: histograms ACCUMULATE into its arrays (as wol_q_traj), series in frame order ----
    if (host) {
        std::vector<unsigned long long> hh((size_t)std::max(nh, 1));
        std::vector<double> all(2 * (size_t)per * W);
        if (nh > 0) HIP_TRY(hipMemcpyAsync(hh.data(), d_hist, sizeof(unsigned long long) * (size_t)nh, hipMemcpyDeviceToHost, c->stream));
        if (mean || var) HIP_TRY(hipMemcpyAsync(all.data(), d_all, sizeof(double) * all.size(), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        if (qhist) for (int k = 0; k < qbins; ++k) qhist[k] += (int64_t)hh[(size_t)k];
        if (dhist) for (int k = 0; k < dbins; ++k) dhist[k] += (int64_t)hh[(size_t)nq + k];
        for (int r = 0; r < W && (mean || var); ++r) {
            int a = 0, b = 0;
            (void)wol_shard_bounds(T_total, r, W, &a, &b);
            for (int t = a; t < b; ++t) {
                if (mean) mean[t] = all[2 * ((size_t)r * per + (t - a))];
                if (var) var[t] = all[2 * ((size_t)r * per + (t - a)) + 1];
            }
        }
    } else {
        if (qhist) hipLaunchKernelGGL(k_add_u64, dim3((qbins + 255) / 256), dim3(256), 0, c->stream, d_hist,
                                      reinterpret_cast<unsigned long long*>(qhist), qbins);
        if (dhist) hipLaunchKernelGGL(k_add_u64, dim3((dbins + 255) / 256), dim3(256), 0, c->stream, d_hist + nq,
                                      reinterpret_cast<unsigned long long*>(dhist), dbins);
        for (int r = 0; r < W && (mean || var); ++r) {
            int a = 0, b = 0;
            (void)wol_shard_bounds(T_total, r, W, &a, &b);
            if (b == a) continue;
            const double* src = d_all + 2 * (size_t)r * per;
            if (mean) HIP_TRY(hipMemcpy2DAsync(mean + a, sizeof(double), src, 2 * sizeof(double), sizeof(double), (size_t)(b - a), hipMemcpyDeviceToDevice, c->stream));
            if (var) HIP_TRY(hipMemcpy2DAsync(var + a, sizeof(double), src + 1, 2 * sizeof(double), sizeof(double), (size_t)(b - a), hipMemcpyDeviceToDevice, c->stream));
        }
    }
    HIP_TRY(hipGetLastError());
    return WOL_OK;
}
}  // namespace

extern "C" {

int wol_shard_bounds(int T_total, int rank, int world, int* t0, int* t1) {
    if (!t0 || !t1 || T_total < 0 || world < 1 || rank < 0 || rank >= world) return fail(WOL_ERR_INVALID, "shard_bounds: bad argument");
    *t0 = (int)(((int64_t)rank * T_total) / world);
    *t1 = (int)(((int64_t)(rank + 1) * T_total) / world);
    return WOL_OK;
}

int wol_comm_unique_id(void* id128) {
    if (!id128) return fail(WOL_ERR_INVALID, "unique id buffer must not be NULL");
    WOL_TRY(rccl_load());
    ncclUniqueId id;
    NCCL_TRY(g_rccl.GetUniqueId(&id));
    memcpy(id128, &id, sizeof id);
    return WOL_OK;
}

int wol_comm_init(wol_ctx* c, const void* id128, int rank, int world, wol_comm** out) {
    WOL_TRY(check_ctx(c));
    if (!id128 || !out || world < 1 || rank < 0 || rank >= world) return fail(WOL_ERR_INVALID, "comm_init: bad argument");
    *out = nullptr;
    WOL_TRY(rccl_load());
    ncclUniqueId id;
    memcpy(&id, id128, sizeof id);
    ncclComm_t nc = nullptr;
    NCCL_TRY(g_rccl.CommInitRank(&nc, world, id, rank));
    wol_comm* m = new (std::nothrow) wol_comm();
    if (!m) { (void)g_rccl.CommDestroy(nc); return fail(WOL_ERR_NOMEM, "out of host memory"); }
    m->comm = nc; m->rank = rank; m->world = world; m->own = true;
    *out = m;
    return WOL_OK;
}

int wol_comm_adopt(void* nccl_comm, int rank, int world, wol_comm** out) {
    if (!nccl_comm || !out || world < 1 || rank < 0 || rank >= world) return fail(WOL_ERR_INVALID, "comm_adopt: bad argument");
    *out = nullptr;
    WOL_TRY(rccl_load());
    wol_comm* m = new (std::nothrow) wol_comm();
    if (!m) return fail(WOL_ERR_NOMEM, "out of host memory");
    m->comm = static_cast<ncclComm_t>(nccl_comm); m->rank = rank; m->world = world; m->own = false;
    *out = m;
    return WOL_OK;
}

int wol_comm_destroy(wol_comm* m) {
    if (!m) return WOL_OK;
    if (m->own && m->comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(m->comm);
    delete m;
    return WOL_OK;
}

int wol_q_traj_sharded(wol_ctx* c, wol_comm* comm, const double* frames, int64_t atom_stride, int64_t frame_stride,
                       int T_local, int T_total, int n, const double* boxes, double lowcut, double highcut, double* mean,
                       double* var, int64_t* qhist, int qbins, int64_t* dhist, int dbins, double dbinwidth, double* q_out,
                       int mem) {
    return q_traj_sharded_impl<double>(c, comm, frames, atom_stride, frame_stride, T_local, T_total, n, boxes, lowcut, highcut,
                                       mean, var, qhist, qbins, dhist, dbins, dbinwidth, q_out, mem);
}

int wol_q_traj_sharded_f32(wol_ctx* c, wol_comm* comm, const float* frames, int64_t atom_stride, int64_t frame_stride,
                           int T_local, int T_total, int n, const double* boxes, double lowcut, double highcut, double* mean,
                           double* var, int64_t* qhist, int qbins, int64_t* dhist, int dbins, double dbinwidth, float* q_out,
                           int mem) {
    return q_traj_sharded_impl<float>(c, comm, frames, atom_stride, frame_stride, T_local, T_total, n, boxes, lowcut, highcut,
                                      mean, var, qhist, qbins, dhist, dbins, dbinwidth, q_out, mem);
}

}  // extern "C"
