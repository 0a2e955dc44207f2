/*
 * nbody_hip.h — C ABI of the MI355X (gfx950) N-body force/integration backend.
 *
 * Drop-in boundary for the hot path of UoB-HPC/stdpar-nbody: the reference has no FFI; its seam is
 * the L3 -> L2 edge where a step driver (run_all_pairs, src/all_pairs.h:52; run_bvh, src/bvh.h:327)
 * issues one blocking stdpar algorithm per phase over the raw-pointer view System::state_t
 * (src/system.h:41-50).  Each entry point below replaces exactly one of those stdpar call sites and
 * takes the same view (`nbody_state`, pointers now in device memory), so a reference maintainer
 * swaps `std::for_each(par_unseq, ...)` for one call (see INTEGRATION.md for the stub).
 *
 * Conventions
 *  - extern "C", plain pointers and sizes, no C++/torch types.  Every function returns 0 on success
 *    and a non-zero code on failure; nbody_last_error() returns the message (thread-local).
 *    No exception crosses the boundary.
 *  - All phase calls are asynchronous on `stream` (a hipStream_t passed as void*, NULL = default
 *    stream).  Ordering between phases is stream order.  nbody_stream_sync() / nbody_download()
 *    are the blocking points.
 *  - dtype: NBODY_F32 | NBODY_F64.  dim: 2 | 3.  Arrays use the reference layout: m is T[sz];
 *    x, v, a, ao are vec<T,D>[...] = D contiguous T per body (src/vec.h:17-19), no padding.
 *  - Sharding (multi-GPU all-pairs): a state owns target bodies [first, first+count).  m and x
 *    always hold ALL sz bodies (sources); v, a, ao hold `count` records, record k = body first+k.
 *    Single GPU: first = 0, count = sz.
 *  - The library never falls back to a CPU path: without a usable HIP device every call fails.
 *  - Devices: a context, a tree and a communicator remember the device they were created on; a phase call runs on
 *    the device of its `stream` (NULL stream: the calling thread's current device); a tree must be used with a stream
 *    of its own device (NBODY_ERR_ARG otherwise).  Every entry point switches to
 *    that device for the duration of the call and restores the caller's, so one host thread can drive several GPUs.
 */
#ifndef NBODY_HIP_H
#define NBODY_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NBODY_F32 0
#define NBODY_F64 1

#define NBODY_OK 0
#define NBODY_ERR_ARG 1     /* bad argument (dtype/dim/null pointer/size) */
#define NBODY_ERR_HIP 2     /* HIP runtime error, message has hipGetErrorString */
#define NBODY_ERR_STATE 3   /* call sequence error (e.g. build_tree before hilbert_sort) */

/* Device-pointer mirror of System<T,N>::state_t (src/system.h:41-50) plus the shard window. */
typedef struct nbody_state {
  void* m;         /* T[sz]            masses of all bodies                         */
  void* x;         /* vec<T,D>[sz]     positions of all bodies                      */
  void* v;         /* vec<T,D>[count]  velocities of owned bodies                   */
  void* a;         /* vec<T,D>[count]  accelerations of owned bodies                */
  void* ao;        /* vec<T,D>[count]  previous-step accelerations of owned bodies  */
  double dt;       /* time step   (converted to T inside, as System::dt is T)       */
  double c;        /* constant G  (System::constant)                                */
  uint32_t sz;     /* total number of bodies (System::size, index_t = uint32_t)     */
  uint32_t first;  /* first owned body                                              */
  uint32_t count;  /* number of owned bodies                                        */
  int32_t dtype;   /* NBODY_F32 | NBODY_F64                                         */
  int32_t dim;     /* 2 | 3                                                         */
  uint32_t tuning; /* K1 launch shape for calls with this view: 0 = the library default, else
                      NBODY_TUNING(split, targets_per_thread, source_path); never changes which
                      pairs are summed (see nbody_all_pairs_configure).  A state filled by hand
                      must zero this field: any other value is refused (NBODY_ERR_ARG)       */
} nbody_state;

/* split in bits 0-3, targets per lane in bits 4-5, source path in bits 6-7; each 0 = auto */
#define NBODY_TUNING(split, targets_per_thread, source_path) \
  ((uint32_t)(((split) & 15) | (((targets_per_thread) & 3) << 4) | (((source_path) & 3) << 6) | 0x100u))

/* ABI version of this header/library pair: major * 1000 + minor.  Bindings should refuse a different major.
 * 2.0: nbody_state.tuning; the collective (nbody_comm_*), shard windows on contexts, per-device guards.
 * 2.1: nbody_bvh_create_on / nbody_octree_create_on (explicit device), nbody_octree_set_walk, nbody_octree_set_build, nbody_octree_set_step_budget,
 *      nbody_bvh_set_launch_order;
 *      a tree used with a stream of another device is refused; nbody_state.tuning is validated (0 or NBODY_TUNING(...)).
 * 2.2: nbody_bvh_read what = 6 and nbody_bvh_opening_thresholds (the opening test as one compare), nbody_all_pairs_pair_rule;
 *      the measured forms that are not shipped (traversal 3 / 4 / 6, octree build 2 / 4) are refused by this library.
 * 2.3: nbody_all_pairs_status; nbody_stream_sync / nbody_download return NBODY_ERR_STATE after a failed K1 chunk hand-off;
 *      nbody_all_pairs_pair_rule decides from the positions' variances, not from their bounding box. */
#define NBODY_HIP_ABI_VERSION 2003
int nbody_abi_version(void);

/* Last error message of the calling thread ("" if none). */
const char* nbody_last_error(void);

/* Library / device identification: writes e.g. "gfx950:sramecc+:xnack-" and the CU count. */
int nbody_device_info(int device, char* arch_out, size_t arch_len, int* cu_count);

/* ---- all-pairs ---------------------------------------------------------------------------------- */

/* K1. Replaces all_pairs_force (src/all_pairs.h:14-27):
 *   a[i] = c * sum_{j != i} m[j] * (x[j] - x[i]) / (pow(|x[j]-x[i]|^2, 3/2) + eps(T))
 * for owned bodies i; sources j = 0..sz-1 staged through LDS tiles. */
int nbody_all_pairs_force(const nbody_state* s, void* stream);

/* K1's in-kernel chunk hand-off, and how it fails.  From 2048 bodies on, the source range is cut into >= 16 chunks over grid.y;
 * the block of chunk y adds its sum into `a` after the block of chunk y - 1 has (a turn word per group of targets), which fixes
 * the rounding order without a scratch array or a second launch.  (Launches whose whole grid is resident at once — up to 2048
 * blocks, i.e. up to 8192 targets — collect the chunks' sums instead: whichever chunk arrives last adds them in chunk order; the same
 * bits, no waiting, nothing that can fail.)  A block waits only for blocks of smaller linear index.  That
 * this terminates rests on an ASSUMPTION about the dispatcher (true of every GPU this library targets, stated nowhere in the
 * ISA): the workgroups of a grid are started in linear index order, so the oldest unfinished block never waits.  The reference's
 * loop (src/all_pairs.h:14-27) cannot return garbage silently, and neither does this: a wave that has waited for its turn past
 * the poll budget (minutes) gives up, the group's rows of `a` end as NaN — never as a finite partial sum — and a sticky
 * per-stream status word is set on the device.  nbody_stream_sync, nbody_download and nbody_all_pairs_status on that stream then
 * return NBODY_ERR_STATE with a message naming the target block, the group and the chunk, until the status is cleared.
 *
 * nbody_all_pairs_status waits for `stream` and writes (out may be NULL)
 *   out[0] = 1 if a hand-off has failed, out[1..3] = target block, target group, chunk of the first failure,
 *   out[4] = polls made by waves that had to wait for their turn, out[5] = number of waves that had to wait
 * accumulated over every K1 launch of this stream since the last clear; clear != 0 zeroes them after the read.  A stream that
 * has never run the chunked K1 reports zeros.  In normal operation nobody waits: out[4] = out[5] = 0. */
int nbody_all_pairs_status(void* stream, uint64_t out[6], int clear);

/* K2. Replaces all_pairs_collapsed_force (src/all_pairs.h:29-50) with its INTENDED semantics
 * (64-bit pair space, all D components; the reference wraps the pair count at 2^32 and drops
 * component 2 — SURVEY §0.5):  a[i] <- (a[i] - ao[i]) + sum_{j != i} (c*m[j]) * (x[j]-x[i]) / dist3.
 * Lanes run along the SOURCE axis (one ordered pair per lane and step), the partial sums of 16 (double: 8) targets are
 * reduced across the wavefront together, one atomic add per target, component and wave (tolerance parity).  Float: a wave
 * owns its 16 targets for a whole chunk of the packed source records (streamed from L2, no LDS); double: (target group x
 * LDS source tile) blocks.  Uses the stream's K1 scratch (packed records) like nbody_all_pairs_force.  Single-GPU only
 * (first = 0, count = sz). */
int nbody_all_pairs_collapsed_force(const nbody_state* s, void* stream);

/* K3. Replaces System::accelerate_step (src/system.h:52-60) for owned bodies:
 *   x += dt*v + ((0.5*dt)*dt)*ao;  v += (0.5*dt)*(a + ao);  ao = a          (bit-exact, no FMA) */
int nbody_accelerate_step(const nbody_state* s, void* stream);

/* K10+K11. Replaces System::calc_energies (src/system.h:62-79): kinetic = 0.5*sum m|v|^2 and potential =
 * -0.5*c*sum_i sum_{j!=i} m_i m_j/(|x_i-x_j| + eps).  Writes one T each to the HOST pointers; blocking.
 * Needs the whole system (first = 0, count = sz). */
int nbody_calc_energies(const nbody_state* s, void* kinetic_out, void* potential_out, void* stream);

/* Tuning knob for K1 (does not change which pairs are summed, only the split of the source range
 * over the waves of a block and hence the rounding order).  split in {0 (auto from sz: 8 for sz >= 65536, else 4), 1, 2,
 * 4, 8 (8: scalar-stream form only)}; targets_per_thread in {0 (auto), 1, 2} (no effect on the result).  The auto split
 * depends on sz only — never on first/count — so results are bitwise independent of how bodies are sharded over GPUs. */
int nbody_all_pairs_configure(int split, int targets_per_thread);
/* The two calls here set the PROCESS-WIDE default (atomic; used by every view whose `tuning` is 0).  A context carries its
 * own choice: nbody_ctx_configure_all_pairs stores it and nbody_ctx_state hands it out in nbody_state.tuning. */
/* How K1 brings a source record to the 64 lanes of a wave (same arithmetic, same order, bitwise the same result):
 * 1 = tiles staged in LDS, read as LDS broadcasts; 2 = records packed once per call and streamed through the scalar
 * unit into SGPRs; 0 = auto (2 once the call has several waves per SIMD, about 65 536 targets; 1 below).  Form 2 keeps a packed-source buffer per calling stream (32 B per body, grow-only):
 * contexts from nbody_create reserve theirs, any other stream gets it on its first call, which must not be recorded. */
int nbody_all_pairs_source_path(int mode);

/* ---- Hilbert BVH (src/bvh.h) ---------------------------------------------------------------------- */

/* Tree + scratch storage; replaces bvh<T,N>::alloc / dealloc (src/bvh.h:147-172).
 * nleafs = bit_ceil(n), nlevels = log2(nleafs), nnodes = 2^nlevels - 1.  n >= 2. */
typedef struct nbody_bvh nbody_bvh;
int  nbody_bvh_create(nbody_bvh** out, int dtype, int dim, uint32_t n);               /* on the current device */
int  nbody_bvh_create_on(nbody_bvh** out, int dtype, int dim, uint32_t n, int device); /* device < 0: the current one */
void nbody_bvh_destroy(nbody_bvh* t);

/* K4. Replaces bounding_box (src/bvh.h:17-22): AABB of all x padded by +-10 eps, always containing
 * the origin.  Result stays on the device for K5; nbody_bvh_get_bounding_box copies it out
 * (blocking): xmin_out/xmax_out are T[dim] on the host. */
int nbody_bvh_bounding_box(nbody_bvh* t, const nbody_state* s, void* stream);
int nbody_bvh_get_bounding_box(nbody_bvh* t, void* xmin_out, void* xmax_out, void* stream);

/* K5+K6. Replaces hilbert_sort (src/bvh.h:25-96): 64-bit Hilbert key per body from the K4 box,
 * stable LSD radix sort of (key, index), then one gather that permutes m, x, v, a, ao IN PLACE
 * (caller-visible, exactly like the reference). Requires first = 0, count = sz. */
int nbody_bvh_hilbert_sort(nbody_bvh* t, const nbody_state* s, void* stream);

/* K7+K8. Replaces bvh::build_tree (src/bvh.h:175-244): leaf-parent level from body pairs, then one
 * launch per level upward (monopoles, AABBs, widths; dead nodes have mass 0 and width 0). */
int nbody_bvh_build_tree(nbody_bvh* t, const nbody_state* s, void* stream);

/* K9. Replaces bvh::compute_force (src/bvh.h:246-324): stackless traversal per owned body with the
 * opening test bw^2 < theta^2 * dist2 evaluated exactly as the reference does (no FMA). */
int nbody_bvh_compute_force(nbody_bvh* t, const nbody_state* s, double theta, void* stream);

/* Test/diagnostic read-back of BVH internals (blocking device->host copies).
 * what: 0 keys u64[n] (pre-sort order) | 1 perm u32[n] (new -> old) | 2 node monopoles T[nnodes][D+1]
 * (x..., mass) | 3 node widths T[nnodes] | 4 node boxes T[nnodes][2D] | 5 traversal counters
 * u32[n][4] {node tests, leaf visits, monopole terms, body terms} (filled by
 * nbody_bvh_compute_force only after nbody_bvh_enable_counters(t, 1)) | 6 opening thresholds T[nnodes]: the largest v
 * with width^2 >= fl(theta^2 * d2) for every d2 <= v, i.e. the reference's test width^2 < theta^2 * d2
 * (src/bvh.h:246-248) is v < d2 bit for bit; written by the build for the angle of the last nbody_bvh_compute_force
 * (0.5, the reference's default, before the first) and rewritten by a traversal that asks for another one. */
int nbody_bvh_read(nbody_bvh* t, int what, void* host_out, size_t bytes, void* stream);
/* The thresholds of what = 6 for given width^2 values, computed on the HOST by the same function the build kernels run (no
 * device needed): out[i] = the largest v such that width2[i] < fl(theta^2 * d2) fails for every d2 <= v; -1 for a negative
 * input (body entries), +inf when no distance accepts.  For the tests that hold this form against src/bvh.h:246-248. */
int nbody_bvh_opening_thresholds(int dtype, const void* width2, double theta, size_t n, void* out);
int nbody_bvh_enable_counters(nbody_bvh* t, int on);
/* K9 scheduling form: 0 = auto (by size), 1 = one independent stackless walk per lane — the reference's loop as is, with its
 * product form of the opening test —, 2 (= 5) = wave-cooperative sweep of the union of the wave's walks in DFS key order, the step
 * program written as ISA, the opening test as one compare against the record's threshold (nbody_bvh_read what = 6).  Both make
 * every body perform the same tests in the same order: results and counters are bitwise identical (the tests hold them equal).
 * Forms that were measured and lost (3 / 4 / 6) exist in the -DNBODY_EXPERIMENTS build only; this library refuses them. */
int nbody_bvh_set_traversal(nbody_bvh* t, int mode);
/* Launch order of the sweep: 0 = work items (groups that straddle a jump of the key order are cut in two and start first),
 * 1 = one block per group in index order.  Bitwise identical results; tests and tuning runs compare the two. */
int nbody_bvh_set_launch_order(nbody_bvh* t, int mode);
uint32_t nbody_bvh_nnodes(const nbody_bvh* t);

/* ---- octree Barnes-Hut (src/octree.h, the reference's default --algorithm) ---------------------------------
 * The reference inserts bodies concurrently under per-node spin locks; only node NUMBERS depend on that order.
 * Here the same spatial tree is built without locks (path keys -> radix sort -> breadth-first split), so results,
 * tree size and visit counts equal the reference's.  Whole system only for the build phases; compute_force
 * honours the shard window.  Errors found on the device (coincident bodies / node pool exhausted) surface in nbody_octree_info. */
typedef struct nbody_octree nbody_octree;
/* octree<T,N>::alloc / dealloc (src/octree.h:42-60); capacity = max(2^dim * n, 1000) nodes (src/system.h:30). */
int  nbody_octree_create(nbody_octree** out, int dtype, int dim, uint32_t n);               /* on the current device */
int  nbody_octree_create_on(nbody_octree** out, int dtype, int dim, uint32_t n, int device); /* device < 0: the current one */
void nbody_octree_destroy(nbody_octree* t);
/* Scheduling form of the walk: 0 = auto, 1 = the compiler-scheduled kernel, 2 = the visit round written as ISA (fails where that
 * form does not exist).  Same tests, same arithmetic, same order: bitwise identical accelerations and counters. */
int  nbody_octree_set_walk(nbody_octree* t, int mode);
/* How the tree is built from the sorted path keys and how the multipole pass is launched:
 *   0 = auto (3);
 *   3 = one pass: every cell follows from the common key prefixes of neighbouring bodies, so all cells are numbered by one prefix
 *       sum and built at once (4 launches whatever the depth), and the multipoles take two or three (rank chunks, then the cells
 *       that span chunk boundaries, in two rounds above 2.6e5 bodies);
 *   1 = breadth-first, one launch per tree level for the build and one for the multipoles (21 + 21 in 3D): the cross-check.
 * 1 numbers the sibling groups breadth-first, 3 in pre-order; the cells, their monopoles, the tree size and every force and
 * counter the walk produces are the same bit for bit.  A tree inserted by one form is not the other's to finish: changing the
 * form clears the insert / tree state.  (The grid-barrier forms 2 and 4 — measured slower on MI355X — exist in the
 * -DNBODY_EXPERIMENTS build only.) */
int  nbody_octree_set_build(nbody_octree* t, int mode);
/* Visit rounds one body's walk may make before it is abandoned and nbody_octree_info reports it (never spin on a damaged
 * tree).  0 = the default: the node pool size, which no walk of a well-formed tree reaches. */
int  nbody_octree_set_step_budget(nbody_octree* t, uint32_t steps);
/* octree::clear (src/octree.h:85-89): readies the tree for the next step. */
int nbody_octree_clear(nbody_octree* t, void* stream);
/* octree::compute_bounds (src/octree.h:93-112): root cube from the scalar min/max over all coordinates, +-1. */
int nbody_octree_compute_bounds(nbody_octree* t, const nbody_state* s, void* stream);
/* octree::insert (src/octree.h:114-181). */
int nbody_octree_insert(nbody_octree* t, const nbody_state* s, void* stream);
/* octree::compute_tree (src/octree.h:183-224): masses and centres of mass, children summed in child order. */
int nbody_octree_compute_tree(nbody_octree* t, void* stream);
/* octree::compute_force (src/octree.h:226-263): a[i] = c * sum over the walk with side/dx < theta.  Every body performs
 * the reference's opening tests and accumulates the reference's terms (the per-body counters are identical); the terms
 * are added per child slot and the 2^dim partial sums combined, so sums differ from the reference's at rounding level. */
int nbody_octree_compute_force(nbody_octree* t, const nbody_state* s, double theta, void* stream);
/* Blocking.  tree_size = next_free_child_group (printed by --print-info, src/octree.h:314), root_mass = m[0].mass()
 * as one T; either may be NULL.  Fails if ANY build since the previous call hit the depth limit or exhausted the node
 * pool (the device-side flag is sticky; this call reports and clears it) — a caller that replays recorded steps checks once
 * at the end of the run. */
int nbody_octree_info(nbody_octree* t, uint32_t* tree_size, void* root_mass, void* stream);
/* Test/diagnostic: per-body {nodes examined, terms accumulated} u32[n][2] of the last compute_force. */
int nbody_octree_enable_counters(nbody_octree* t, int on);
int nbody_octree_read_counters(nbody_octree* t, uint32_t* host_out, size_t bytes, void* stream);

/* ---- owning context (device mirrors of a host System), used by the C++ CLI host ------------------ */

typedef struct nbody_ctx nbody_ctx;
/* Allocates device arrays for n bodies on `device` and a private stream. */
int  nbody_create(nbody_ctx** out, int dtype, int dim, uint32_t n, int device);
void nbody_destroy(nbody_ctx* ctx);
/* Host -> device copy of the System arrays (T[n], vec<T,D>[n] x4) and the scalars dt, c. Blocking. */
int  nbody_upload(nbody_ctx* ctx, const void* m, const void* x, const void* v, const void* a, const void* ao, double dt,
                  double c);
/* Device -> host; any pointer may be NULL to skip that array.  m too: bvh permutes it. Blocking. */
int  nbody_download(nbody_ctx* ctx, void* m, void* x, void* v, void* a, void* ao);
/* The device view and stream of the context, to pass to the phase calls above. */
int  nbody_ctx_state(nbody_ctx* ctx, nbody_state* out);
void* nbody_ctx_stream(nbody_ctx* ctx);
/* Blocks until all work queued on `stream` has completed (so wall-clock phase timers are honest). */
int  nbody_stream_sync(void* stream);
/* K1 launch shape of THIS context (0 = auto for each; see nbody_all_pairs_configure / nbody_all_pairs_source_path). */
int  nbody_ctx_configure_all_pairs(nbody_ctx* ctx, int split, int targets_per_thread, int source_path);
/* Multi-GPU all-pairs: this context owns target bodies [first, first+count) of the n it was created for.  m and x stay
 * whole (sources); nbody_ctx_state then returns the window with v/a/ao pointing at the owned rows, and nbody_download
 * writes only the owned rows of x, v, a, ao (at their place in the full-size host arrays) plus all of m. */
int  nbody_ctx_set_shard(nbody_ctx* ctx, uint32_t first, uint32_t count);
/* What K1 will launch for this view, e.g. "all_pairs_force_sgpr_kernel<double,3,R=2,JS=8> tile=512 pair=far3/near2"
 * (bench.py stamps its profiles with it). */
int  nbody_all_pairs_describe(const nbody_state* s, char* out, size_t len);
/* Which per-pair rounding form K1 (sz >= 32768) takes for this state — a property of ALL sz positions, the same on every rank
 * and for every shard window: *sparse_out = 1 when the volume (area in 2D; through *volume_out if not NULL) of the uniformly
 * filled box that has the positions' variances, prod_k sqrt(12 var_k), is at least 1.7e5 (6.4e4 in 2D); pairs at r^2 >= 4 then
 * drop the eps term of m / (r^3 + eps), which is below an eighth of an ulp there (float: they take m r^-3 from the reciprocal
 * square root alone).  0: the dense rule (always below 32768 bodies).  The moments are summed in a fixed order (same bits on
 * every rank and in every run).  Until ABI 2.2 the volume was the bounding box's, which a single escaping body inflates at
 * will: config 2 as written took the sparse rule from its 36th step on with every batch of pairs holding a close one (+ 8.5 %).
 * Either form is within 2.5 ulp per term, but a system that spreads out moves from one to the other between two steps; tests
 * and bitwise A/B runs query the rule in force with this call.  Blocking (one reduction + a 16-byte copy). */
int  nbody_all_pairs_pair_rule(const nbody_state* s, void* stream, int* sparse_out, double* volume_out);

/* ---- the collective: per-step all-gather of position shards (multi-GPU all-pairs; no reference counterpart) ------
 * RCCL over xGMI.  Partition fixed by the ABI: rank r of W owns bodies [sz*r/W, sz*(r+1)/W) (nbody_shard_range).
 * Every rank holds all of x; nbody_allgather_positions fills in the other ranks' rows in place, asynchronously on
 * `stream` (stream-ordered after the K3 that moved the owned rows): one in-place ncclAllGather when W divides sz,
 * else one grouped ncclSend/ncclRecv per peer.  RCCL is loaded on first use (dlopen), never at library load.
 *  - one process per GPU: rank 0 calls nbody_comm_get_unique_id, the launcher hands the NBODY_COMM_ID_BYTES to every
 *    rank (bench.py: torch.distributed broadcast), each rank calls nbody_comm_create;
 *  - one process, several GPUs (the CLI's --gpus N): nbody_comm_create_all (ncclCommInitAll); the host thread brackets
 *    the per-device nbody_allgather_positions calls of a step with nbody_comm_group_begin/end. */
typedef struct nbody_comm nbody_comm;
#define NBODY_COMM_ID_BYTES 128
int  nbody_comm_get_unique_id(void* id_out);
int  nbody_comm_create(nbody_comm** out, int world, int rank, const void* unique_id, int device);
int  nbody_comm_create_all(nbody_comm** out /* [ndev] */, int ndev, const int* devices /* NULL = 0..ndev-1 */);
void nbody_comm_destroy(nbody_comm* comm);
int  nbody_comm_world(const nbody_comm* comm);
int  nbody_comm_rank(const nbody_comm* comm);
int  nbody_comm_rccl_version(void); /* ncclGetVersion code, 0 if RCCL cannot be loaded */
void nbody_shard_range(uint32_t sz, int world, int rank, uint32_t* first, uint32_t* count);
int  nbody_comm_group_begin(void);
int  nbody_comm_group_end(void);
int  nbody_allgather_positions(nbody_comm* comm, const nbody_state* s, void* stream);

/* ---- step graphs ----------------------------------------------------------------------------------------
 * A simulation step is a fixed sequence of phase calls (5 launches for all-pairs, ~40 for bvh).  Between
 * nbody_graph_begin and nbody_graph_end the phase calls made on `stream` are recorded instead of executed
 * (HIP stream capture); nbody_graph_launch replays the whole step with one submission.  Only the asynchronous
 * phase calls may be recorded (no upload/download/read/get/sync/calc_energies inside a capture). */
typedef struct nbody_graph nbody_graph;
int  nbody_graph_begin(void* stream);
int  nbody_graph_end(void* stream, nbody_graph** out);
int  nbody_graph_launch(nbody_graph* g, void* stream);
void nbody_graph_destroy(nbody_graph* g);

#ifdef __cplusplus
}
#endif
#endif /* NBODY_HIP_H */
