/*
 * wol.h -- C ABI of libwol.so, the MI355X (gfx950) implementation of WaterOrderLib's per-frame
 * water order-parameter hot path.  This is the drop-in boundary: every entry point below is what
 * the reference's FFI for this path binds (the f2py module `waterlib`, generated from
 * fortran/waterlib.f90 by fortran/buildWrappers.sh:3-5) or what its per-frame Python loop
 * (structureLibs/water_properties.py) would call once the per-water work is moved off the host.
 * The reference-side binding a maintainer adds is shown in INTEGRATION.md.
 *
 * Conventions
 *   - plain C types only; every function returns an int status (WOL_OK == 0, <0 = error) and
 *     never aborts the process (the reference's Fortran `stop`s, e.g. waterlib.f90:22-25);
 *     wol_last_error() returns the message of the calling thread's last failure.
 *   - positions are row-major [n][3] doubles (what the Python callers hold: `pos[watInds]`,
 *     orderParam_lib.py:1462); a row stride (in doubles, >= 3) lets the library read O rows
 *     straight out of an interleaved [natoms][3] frame (stride = 3*lenWat, TrajObject.py:35-52)
 *     without the host-side gather.
 *   - `box` is the three orthorhombic box lengths (frame.box.values[:3], orderParam_lib.py:1461)
 *     and is ALWAYS a host pointer.  BoxL < 0 marks a non-periodic axis (iBoxL = 0,
 *     waterlib.f90:41).
 *   - `mem` says where the bulk arrays (positions and outputs) live: WOL_MEM_HOST (the library
 *     stages them) or WOL_MEM_DEVICE (HIP device pointers, e.g. torch tensors' data_ptr(); work is
 *     enqueued on the context's stream and the call returns without synchronising).
 *   - masks are int32, Fortran order, exactly what f2py returns for a `logical` array.
 *   - a context is bound to one device and is not thread-safe: one context per host thread/GPU.
 */
#ifndef WOL_H
#define WOL_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define WOL_OK 0
#define WOL_ERR_INVALID (-1)  /* bad argument (shape, stride, NULL) */
#define WOL_ERR_HIP (-2)      /* a HIP runtime call failed */
#define WOL_ERR_NOMEM (-3)    /* device or host allocation failed */
#define WOL_ERR_NODEVICE (-4) /* no usable GPU */
#define WOL_ERR_PEER (-5)     /* frame-sharded calls: another rank failed in its local part (this rank's results are not written) */

#define WOL_MEM_HOST 0
#define WOL_MEM_DEVICE 1

/* storage type of the caller's real arrays in the *_t entry points (arithmetic is always the reference's fp64:
 * floats are promoted exactly on the way in and results rounded once on the way out) */
#define WOL_F64 0
#define WOL_F32 1

typedef struct wol_ctx wol_ctx;
typedef struct wol_comm wol_comm;   /* RCCL communicator of a frame-sharded job (one rank per GPU) */

int wol_version(void);
const char* wol_last_error(void);

/* context: device ordinal; owns the stream (unless one is set), workspace and staging buffers */
int wol_ctx_create(int device, wol_ctx** out);
int wol_ctx_destroy(wol_ctx* ctx);
/* run on the caller's hipStream_t (e.g. torch.cuda.current_stream().cuda_stream) instead of the
 * context's own stream; NULL selects HIP's default (null) stream, which is what torch's default
 * stream is.  Cheap when the stream is unchanged. */
int wol_ctx_set_stream(wol_ctx* ctx, void* hip_stream);
int wol_ctx_synchronize(wol_ctx* ctx);
/* tunables: "cell_width" (A, default 4.5), "frames_per_batch" (0 = automatic), "path" (0 auto, 1 force
 * all-pairs, 2 force exact cell kernel, 3 force filtered cell kernel), "search_rows" (1 = row-merged proposal
 * kernel, default; 0 = 3-D tile proposal kernel), "lds_sort" (1 = streaming / LDS sorts when they apply,
 * default; 0 = global-atomic form), "sort_parts" (blocks per frame of the streaming sort's first level, 0 =
 * automatic), "overlap" (1 = sort the next batch on a second stream beside the current batch's search;
 * default 0: measured +0-2 % once the proposal kernel draws its tiles dynamically), "stages" (pipe depth with "overlap":
 * 2 = sort | proposal + certification (default), 3 = the certification of the batch before on a third stream as well),
 * "search_margin" (blocks per CU the proposal kernel leaves free for such an overlapped sort), "seg_sort" (first sort level: 1 = single-pass segment form for frames of 24 576 waters
 * and more -- split over several blocks per frame for big frames in small batches -- default; 2 = whenever it applies; 0 = row-bucket form), "dyn_tiles" (1 = the
 * proposal kernel's blocks draw their tiles from per-XCD counters, default: -8 % kernel time against the fixed order, 0), "inputs_resident" (1 = the caller guarantees that device input buffers are complete in
 * memory when a call is made, not merely enqueued earlier on the stream, so the sort of one call may overlap
 * the previous call's kernels; default 0), "time_kernels" (0/1: HIP-event timing of the dominant kernel,
 * read back with wol_ctx_kernel_time), "rows_block" (0 = the proposal kernel's instantiation is picked per batch by lane
 * fill, default; 512 / 768 = force that k_search_rows instantiation: tests and tuning), "check_proposals" (diagnostic, 0/1:
 * the proposal arrays are poisoned before the row-merged proposal kernel and a counting kernel behind it checks that the
 * kernel wrote a key for every water it was given -- counters "proposals_written", "proposals_expected", "proposals_bad";
 * switching it on or off restarts the counters), "host_pack" (what host-memory frames cross the link as: 0 = whole
 * interleaved frames, one linear copy per batch; 1 = the O rows only, packed by host threads into pinned staging -- what
 * the reference's caller does with pos[watInds], orderParam_lib.py:1462; 2 = the same, as float32 whenever every coordinate
 * of the batch is exactly a float32 -- bit-identical results at half the bytes; default 2), "host_threads" (packing
 * threads, 0 = min(16, hardware threads)), "rec32" (1 = float32 frames of the trajectory q calls keep their values as
 * 16-byte (x, y, z, index) float records through both sort levels and in the certification kernel's gathers, promoted
 * exactly where the fp64 arithmetic starts -- bit-identical results, half the sort's bytes; default; 0 = 32-byte double
 * records as for float64 frames), "idx_sort" (1 = float64 frames of the trajectory q calls go through the sort as 16-byte
 * index records and the certification reads the coordinates from the caller's frames by original index -- bit-identical
 * results, 44 % fewer bytes through the sort, but the certification's gathers lose their locality: measured 8-15 % slower
 * per step, default 0) */
int wol_ctx_set_option(wol_ctx* ctx, const char* key, double value);
/* accumulated HIP-event time (ms) and launch count of the dominant (neighbour search + q) kernel
 * since the last reset; needs option time_kernels=1.  Synchronises the stream. */
int wol_ctx_kernel_time(wol_ctx* ctx, double* ms_total, int64_t* launches, int reset);

/* diagnostic counters of the LAST q call on this context (synchronises the stream):
 * "slow_waters" = waters the fp32 filter could not certify and the exact search finished;
 * "density_level" = tile-shape level the device picked for frame 0 (0 = box-average density);
 * "tile_sx" / "tile_ry" / "tiles_per_frame" = the proposal kernel's tiles of frame 0 at that level (cells along x, cell rows
 * along y); "cells_x" / "cells_y" / "cells_z" = frame 0's grid;
 * "rows_block" = threads per block of the k_search_rows instantiation that ran (0: another proposal kernel);
 * "h2d_bytes" = bytes the last host-memory trajectory call sent over the host link;
 * "record_bytes" = size of a sorted record in the last q batch (32; 16 for float32 frames, option "rec32"; 4 = index
 * records, option "idx_sort");
 * "proposals_written" / "proposals_expected" / "proposals_bad": see option "check_proposals" (accumulated, not per call) */
int wol_ctx_get_counter(wol_ctx* ctx, const char* key, int64_t* value);

/* ---- level 1: the f2py `waterlib` routines on the path (host pointers) -------------------- */
/* waterlib.f90:32-47 reimage(Pos,RefPos,BoxL) -> ReimagedPos(NPos,3); pos/out row-major [n][3] */
int wol_reimage(wol_ctx* ctx, const double* pos, int n, const double* refpos, const double* box,
                double* out);
/* waterlib.f90:710-743 nearNeighbors -> logical(NsubPos,NPos) as int32 Fortran order */
int wol_near_neighbors(wol_ctx* ctx, const double* sub, int ns, const double* pos, int n,
                       const double* box, double lowcut, double highcut, int32_t* mask);
/* waterlib.f90:830-862 allNearNeighbors -> logical(NPos,NPos) */
int wol_all_near_neighbors(wol_ctx* ctx, const double* pos, int n, const double* box, double lowcut,
                           double highcut, int32_t* mask);
/* waterlib.f90:867-895 tetraCosAng -> allAngs(n,n), degrees, symmetric; the diagonal (left
 * uninitialised by the reference) is written as 0 */
int wol_tetra_cos_ang(wol_ctx* ctx, const double* refpos, const double* neigh, int n,
                      const double* box, double* allangs);
/* waterlib.f90:900-918 lsiDists -> dist(n) */
int wol_lsi_dists(wol_ctx* ctx, const double* refpos, const double* neigh, int n, const double* box,
                  double* dist);

/* ---- level 2: one frame of water_properties.py ------------------------------------------- */
/* getOrderParamq (water_properties.py:344-391), fused: minimum-image neighbour search, 4-NN
 * selection, six angles, q.  sub == NULL means sub is pos (the allnearneighbors branch, wp:363).
 * Outputs (q required, others nullable):
 *   q[ns]; nn_idx[ns][4] global index into pos, -1 padded; nn_dist[ns][4] ascending, NaN padded;
 *   ang_deg[ns][6] the six angles that enter q (180 padded as wp:379-384, NaN if no neighbour). */
int wol_q_frame(wol_ctx* ctx, const double* pos, int64_t pos_stride, int n, const double* sub,
                int64_t sub_stride, int ns, const double* box, double lowcut, double highcut,
                double* q, int32_t* nn_idx, double* nn_dist, double* ang_deg, int mem);

/* The same with the caller's reals stored as `dtype` (WOL_F64 / WOL_F32) in `mem` (host or device): pos, sub, q,
 * nn_dist and ang_deg are arrays of that type; nn_idx is int32 in the same memory space. */
int wol_q_frame_t(wol_ctx* ctx, const void* pos, int64_t pos_stride, int n, const void* sub, int64_t sub_stride,
                  int ns, const double* box, double lowcut, double highcut, void* q, int32_t* nn_idx,
                  void* nn_dist, void* ang_deg, int dtype, int mem);

/* getCosAngs (water_properties.py:210-250): all k(k-1)/2 angles per water over the shell
 * lowcut < d <= highcut, flattened in water order.  Two calls: with ang == NULL it fills
 * num_angs[ns] (= k per water, as a double like the reference's np.zeros) and *total; then with a
 * buffer of *total doubles.  Host pointers. */
int wol_cos_angs(wol_ctx* ctx, const double* pos, int n, const double* sub, int ns,
                 const double* box, double lowcut, double highcut, double* ang, double* num_angs,
                 int64_t* total);

/* typed / device-memory form: pos, sub, ang, num_angs are `dtype` arrays in `mem`; *total is always a host value
 * (it sizes the second call's buffer), so the call synchronises the stream */
int wol_cos_angs_t(wol_ctx* ctx, const void* pos, int n, const void* sub, int ns, const double* box,
                   double lowcut, double highcut, void* ang, void* num_angs, int64_t* total, int dtype, int mem);

/* getLSI (water_properties.py:252-311): lsi_full[ns] = LSI of water i or NaN where the reference
 * skips the water; num_lsi[ns] as wp:306; *nvalid = len(lsiVals).  Host pointers. */
int wol_lsi_frame(wol_ctx* ctx, const double* pos, int n, const double* sub, int ns,
                  const double* box, double lowcut, double highcut, double* lsi_full,
                  double* num_lsi, int64_t* nvalid);

/* typed / device-memory form; *nvalid is a host value (the call synchronises the stream) */
int wol_lsi_frame_t(wol_ctx* ctx, const void* pos, int n, const void* sub, int ns, const double* box,
                    double lowcut, double highcut, void* lsi_full, void* num_lsi, int64_t* nvalid, int dtype, int mem);

/* ---- "next" row f-1: Willard-Chandler density field ---------------------------- */
/* waterlib.f90:1286-1341 WillardDensityField (called by surface_library.densityGrid, :170-210):
 * truncated-shifted Gaussian density (sigma = smoothlen, cut at 3 sigma) and its unit gradient at
 * the nx*ny*nz grid points.  densvals (nx,ny,nz) and densnorms (nx,ny,nz,3) are Fortran-ordered, as
 * f2py returns them; points with no water in range get NaN normals (0/0), as the reference.  Host
 * pointers. */
int wol_willard_density_field(wol_ctx* ctx, const double* pos, int n, const double* gridx, int nx,
                              const double* gridy, int ny, const double* gridz, int nz,
                              const double* box, double smoothlen, double* densvals, double* densnorms);
/* waterlib.f90:1351-1398 WillardDensityPoints: the same at npts arbitrary points; densnorms is
 * row-major [npts][3]. */
int wol_willard_density_points(wol_ctx* ctx, const double* pos, int n, const double* pts, int npts,
                               const double* box, double smoothlen, double* densvals,
                               double* densnorms);

/* ---- "next" row f-2: water orientation -------------------------------------------- */
/* waterlib.f90:973-1011 watOrient (called by water_properties.waterOrientation / waterOrientationBinZ,
 * :578-638): per water the angle (degrees) between a reference direction and (i) its dipole = sum of
 * the two minimum-imaged O-H vectors, (ii) the normal OH1 x OH2 of its molecular plane.  opos [n][3],
 * hpos [2n][3] (hydrogens of water i in rows 2i, 2i+1), host pointers. */
int wol_wat_orient(wol_ctx* ctx, const double* opos, const double* hpos, int n, const double* refvec,
                   const double* box, double* angdip, double* angplane);
/* waterlib.f90:1019-1044 watOHvec: unit dipole vectors, row-major [n][3]. */
int wol_wat_oh_vec(wol_ctx* ctx, const double* opos, const double* hpos, int n, const double* box,
                   double* normdip);

/* ---- "next" row f-3: radial distribution functions ----------------------------- */
/* waterlib.f90:316-357 RadialDistSame (called per frame by orderParam_lib.rdfCalc, :624-630): g(r) of one
 * set, pairs counted once, bins ((k-1)w, k w], normalised as the reference does (NPos * BulkDens *
 * (4./3.) * pi * w^3 * (k^3 - (k-1)^3), single-precision 4./3., pi = 3.141592653589).  Host pointers. */
int wol_radial_dist_same(wol_ctx* ctx, const double* pos, int n, double binwidth, int totbins,
                         double bulkdens, const double* box, double* rdf);
/* waterlib.f90:193-231 RadialDist: atoms of pos2 around atoms of pos1 (normalised by n1). */
int wol_radial_dist(wol_ctx* ctx, const double* pos1, int n1, const double* pos2, int n2,
                    double binwidth, int totbins, double bulkdens, const double* box, double* rdf);
/* waterlib.f90:363-389 PairDistanceHistogram for Dim = 3: raw counts (as doubles), zero distances skipped. */
int wol_pair_distance_histogram(wol_ctx* ctx, const double* pos1, int n1, const double* pos2, int n2,
                                double binwidth, int totbins, const double* box, double* hist);

/* ---- "next" row f-4: hydrogen bonds ------------------------------------------------ */
/* waterlib.f90:1156-1210 generalHbonds (called four times per frame by orderParam_lib.getBoundWrap, :519-545):
 * bond (nacc, ndon) Fortran-order int32, 1 where acceptor i and donor j (heavy atom don[j], hydrogen
 * donh[j]) are hydrogen bonded: 1.0E-2 < |D-A|^2 <= distCut^2 and angle(A-H, D-H) >= angCut degrees. */
int wol_general_hbonds(wol_ctx* ctx, const double* acc, int nacc, const double* don, const double* donh,
                       int ndon, const double* box, double distcut, double angcut, int32_t* bond);
/* The same bonds as a list (acceptor, donor), ascending -- np.where(bondBool) -- through the cell list, for
 * frames whose mask would not fit.  *count = bonds found; if it exceeds capacity only the first
 * `capacity` (in no particular selection) are returned: call again with a larger buffer. */
int wol_general_hbonds_list(wol_ctx* ctx, const double* acc, int nacc, const double* don,
                            const double* donh, int ndon, const double* box, double distcut,
                            double angcut, int64_t capacity, int32_t* acc_idx, int32_t* don_idx,
                            int64_t* count);

/* ---- level 3: the tetOrderCalc frame loop (orderParam_lib.py:1458-1495) ------------------- */
/* T frames of n waters each; water w of frame t is at frames + t*frame_stride + w*atom_stride
 * (strides in doubles).  boxes is host [T][3].  Per frame: q for every water (sub == pos branch),
 * np.mean/np.var (mean[T], var[T]); over all frames: np.histogram(q, qbins, [0,1]) counts added
 * into qhist[qbins] and the sorted 4-NN O-O distances binned by the reference's pair-histogram
 * rule nbin = ceiling(d/binwidth) (waterlib.f90:219-222) added into dhist[dbins].  Histograms are
 * ACCUMULATED (caller zeroes them) so frame shards and ranks can be summed.  q_out (nullable)
 * receives q[T][n].  Any of mean/var/qhist/dhist/q_out may be NULL. */
int wol_q_traj(wol_ctx* ctx, const double* frames, int64_t atom_stride, int64_t frame_stride, int T,
               int n, const double* boxes, double lowcut, double highcut, double* mean, double* var,
               int64_t* qhist, int qbins, int64_t* dhist, int dbins, double dbinwidth,
               double* q_out, int mem);

/* same with float32 storage (BASELINE config 5): coordinates are promoted to double exactly, the
 * arithmetic is the fp64 path's, q_out is rounded to float */
int wol_q_traj_f32(wol_ctx* ctx, const float* frames, int64_t atom_stride, int64_t frame_stride,
                   int T, int n, const double* boxes, double lowcut, double highcut, double* mean,
                   double* var, int64_t* qhist, int qbins, int64_t* dhist, int dbins,
                   double dbinwidth, float* q_out, int mem);

/* ---- level 3, continued: populations, and the other two frame loops of the path --------------------------------
 * tetOrderCalc, threeBodyCalc and lsiCalc all take `subInds`: per frame t and population j a list of atoms whose
 * positions pos[subInds[t][j]] are handed to the per-frame routine as subPos (orderParam_lib.py:1464-1467, 1318-1322,
 * 1624-1628).  Here a population is given the same way, by POSITION: pop_pos holds the query points of all frames and
 * populations back to back ([total][3] reals of `dtype` in `mem`), points [pop_off[t * npops + j], pop_off[t * npops +
 * j + 1]) belong to frame t, population j (pop_off: HOST array of T * npops + 1 offsets, pop_off[0] = 0).  npops = 0
 * (pop_pos / pop_off NULL) = no populations.  A point that coincides with a water does not see itself (d^2 > lowcut^2),
 * exactly as in the reference's subPos != Pos branch, so a population of water oxygens gets those waters' values.
 * Results are laid out by SEGMENT: seg = t * (npops + 1) + j with j = 0 for all waters of frame t and j >= 1 for
 * population j - 1.  `frames`, `pop_pos` and the real-valued per-water / per-point outputs are `dtype` arrays
 * (WOL_F64 / WOL_F32: float storage is promoted exactly, the arithmetic is the fp64 path's) in `mem`; integer and
 * double outputs live in `mem` too; boxes and pop_off are host arrays.  Frames are batched through HBM as in wol_q_traj
 * (host frames: double-buffered staging, option "host_pack"). */

/* tetOrderCalc with populations (orderParam_lib.py:1458-1495): everything wol_q_traj returns for population 0 (all
 * waters) plus, per frame and population, np.mean / np.var of the population's q (pop_mean / pop_var [T][npops]; NaN
 * for an empty population, as numpy), np.histogram(q, qbins, [0, 1]) of every population ACCUMULATED into
 * pop_qhist[npops][qbins], and optionally every point's q (pop_q[total]).  Any output may be NULL. */
int wol_q_traj_pops(wol_ctx* ctx, const void* frames, int64_t atom_stride, int64_t frame_stride, int T, int n,
                    const double* boxes, double lowcut, double highcut, double* mean, double* var, int64_t* qhist,
                    int qbins, int64_t* dhist, int dbins, double dbinwidth, void* q_out, const void* pop_pos,
                    const int64_t* pop_off, int npops, double* pop_mean, double* pop_var, int64_t* pop_qhist,
                    void* pop_q, int dtype, int mem);

/* threeBodyCalc's frame loop (orderParam_lib.py:1312-1353): per segment what wp.tetrahedralMetrics
 * (water_properties.py:314-342) keeps of wp.getCosAngs's angles (:210-250; shell lowcut < d <= highcut, reference
 * default 3.413): ang_hist[seg][nbins] = np.histogram(angles, nbins, [0, 180]) (written, not accumulated: sum over
 * frames for the pooled distribution the reference writes to 3bDistribution_j.txt; the per-frame histogram gives the
 * frame's entropy); counts[seg] = {number of angles, number of angles in [100, 120]}; cos_sums[seg] = {sum,
 * sum of squares} of cos(angle * pi / 180) over the angles in [100, 120] (so fracTet = counts[1] / counts[0], avgCos =
 * cos_sums[0] / counts[1], varCos = cos_sums[1] / counts[1] - avgCos^2).  The sums are accumulated as exact fixed-point
 * integers: results do not depend on batching, scheduling or the order of the waters. */
int wol_cos_angs_traj(wol_ctx* ctx, const void* frames, int64_t atom_stride, int64_t frame_stride, int T, int n,
                      const double* boxes, double lowcut, double highcut, const void* pop_pos, const int64_t* pop_off,
                      int npops, int nbins, int64_t* ang_hist, int64_t* counts, double* cos_sums, int dtype, int mem);

/* lsiCalc's frame loop (orderParam_lib.py:1618-1640): wp.getLSI (water_properties.py:252-311; shell lowcut < d <=
 * highcut, "next" neighbour in (highcut, highcut + 3.7] chosen by raw distance as the reference does) per segment:
 * counts[seg] = len(lsiVals) (waters the reference skips are not counted), mean[seg] / var[seg] = np.mean / np.var of
 * them (NaN for none), lsi_hist[npops + 1][nbins] = np.histogram(lsiVals, nbins, [0, hist_max]) per population over
 * all frames, ACCUMULATED (the reference: 500 bins over [0, 0.3]).  Optional: lsi_out[T][n] = LSI of every water in the
 * caller's water order (NaN where the reference skips it), pop_lsi[total] likewise for the population points.
 * The kernel keeps a water's shell in LDS rows sized for twice what the mean density predicts; denser shells (clusters) are
 * finished from rows in global memory, however many waters of a batch that takes.  A shell of more than 1022 waters is
 * reported as WOL_ERR_INVALID: use wol_lsi_frame for such inputs.  Both shell trajectory calls synchronise the stream
 * once, at the end, to report that condition. */
int wol_lsi_traj(wol_ctx* ctx, const void* frames, int64_t atom_stride, int64_t frame_stride, int T, int n,
                 const double* boxes, double lowcut, double highcut, const void* pop_pos, const int64_t* pop_off,
                 int npops, int nbins, double hist_max, int64_t* lsi_hist, int64_t* counts, double* mean, double* var,
                 void* lsi_out, void* pop_lsi, int dtype, int mem);

/* rdfCalc's frame loop for the water-water g(r) (orderParam_lib.py:624-630 calls wl.radialdistsame once per frame): per frame
 * the pair counts of RadialDistSame (counts[T][totbins], bins ((k-1)w, k w], pairs once) and / or g(r) normalised exactly as
 * wol_radial_dist_same does (rdf[T][totbins]; the caller sums frames in its own order, as the reference does).  Either
 * output may be NULL.  Synchronises the stream (the normalisation is redone on the host from the integer counts). */
int wol_radial_dist_same_traj(wol_ctx* ctx, const void* frames, int64_t atom_stride, int64_t frame_stride, int T, int n,
                              const double* boxes, double binwidth, int totbins, double bulkdens, double* rdf,
                              int64_t* counts, int dtype, int mem);

/* ---- multi-GPU: frame shards, one exchange step ------------------------------------------
 * The frame loop of tetOrderCalc has no dependence between frames (orderParam_lib.py:1458-1473): rank r of R (one
 * process per GPU) owns the contiguous block of frames given by wol_shard_bounds and runs the one-GPU pipeline on it;
 * the job ends with ONE exchange step over RCCL -- all-reduce (sum) of the int64 histograms, all-gather of the per-frame
 * (mean, var) series.  Integer sums do not depend on the order of reduction: every rank ends up with exactly the
 * one-GPU result. */
/* frames [*t0, *t1) = [rank*T/world, (rank+1)*T/world) belong to `rank` */
int wol_shard_bounds(int T_total, int rank, int world, int* t0, int* t1);
/* Communicator.  Rank 0 calls wol_comm_unique_id (fills 128 bytes), the caller hands those bytes to every rank by
 * whatever it has (MPI_Bcast, a file, a torch.distributed store), every rank calls wol_comm_init with its context
 * (whose device becomes the rank's GPU).  wol_comm_adopt wraps an ncclComm_t the caller already owns.  RCCL is looked
 * up at run time (the copy already mapped into the process, else librccl.so.1 on the loader path). */
int wol_comm_unique_id(void* id128);
int wol_comm_init(wol_ctx* ctx, const void* id128, int rank, int world, wol_comm** out);
int wol_comm_adopt(void* nccl_comm, int rank, int world, wol_comm** out);
int wol_comm_destroy(wol_comm* comm);
/* wol_q_traj on this rank's T_local frames (frames / boxes / q_out are the LOCAL block) followed by the exchange step:
 * qhist / dhist receive (accumulate, as wol_q_traj) the job-wide counts, mean / var [T_total] the series of all frames
 * in trajectory order, on every rank.  Collective: every rank of the communicator must call it.
 * Errors are agreed on: a rank whose local part fails (bad arguments, a failed copy or launch, out of memory in the pipeline)
 * still takes part in the exchange with zeros and a raised status word that rides on the histogram all-reduce, so no
 * rank is left waiting inside RCCL; that rank returns its own error, every other rank WOL_ERR_PEER, and nobody's output
 * arrays are written.  Only a rank that cannot reach the exchange at all (no device, NULL communicator, no memory for
 * the KB-sized exchange buffer) returns early -- abort the communicator then, as for any collective.  The status word is
 * read on the host, so the call synchronises the context's stream once in either memory mode.
 * The world > 1 path has not yet run on hardware (no multi-GPU box was available to this build): covered by a
 * one-rank RCCL test and a two-rank test that is skipped on one-GPU boxes. */
int wol_q_traj_sharded(wol_ctx* ctx, wol_comm* comm, const double* frames, int64_t atom_stride, int64_t frame_stride,
                       int T_local, int T_total, int n, const double* boxes, double lowcut, double highcut,
                       double* mean, double* var, int64_t* qhist, int qbins, int64_t* dhist, int dbins,
                       double dbinwidth, double* q_out, int mem);
int wol_q_traj_sharded_f32(wol_ctx* ctx, wol_comm* comm, const float* frames, int64_t atom_stride, int64_t frame_stride,
                           int T_local, int T_total, int n, const double* boxes, double lowcut, double highcut,
                           double* mean, double* var, int64_t* qhist, int qbins, int64_t* dhist, int dbins,
                           double dbinwidth, float* q_out, int mem);

#ifdef __cplusplus
}
#endif
#endif /* WOL_H */
