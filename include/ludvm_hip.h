/*
 * ludvm_hip.h -- C ABI of libludvm_hip.so, the MI355X (gfx950) all-pairs vortex-induction engine
 * that sits behind the LUDVM Python class.
 *
 * The reference (jcatalang/LUDVM) has no FFI: its "operator boundary" for this path is the Python
 * method surface of class LUDVM.  Each entry point below cites the reference code it replaces
 * (file:line into LUDVM.py).  The Python host (ludvm_amd/_ffi.py, ctypes) binds exactly these
 * symbols; INTEGRATION.md shows the stub a reference maintainer would add.
 *
 * Conventions
 *   - every function returns an int status: 0 = LUDVM_OK, otherwise one of LUDVM_E_*; nothing
 *     throws across the ABI and nothing calls exit();
 *   - ludvm_last_error(ctx) returns a human-readable message for the last failure on that context
 *     (pointer owned by the context, valid until the next call on it);
 *   - plain pointers and sizes only; "host" pointers are ordinary process memory, "dev" pointers
 *     are HIP device memory on the context's device (e.g. a torch tensor's data_ptr());
 *   - host-pointer entry points are synchronous (results are in the output arrays on return);
 *     dev-pointer entry points are asynchronous on the context's stream (ludvm_set_stream /
 *     ludvm_synchronize);
 *   - a context is not thread-safe: one host thread at a time per context (the reference is
 *     single-threaded); distinct contexts are independent;
 *   - coordinates follow the reference: x (horizontal), z (vertical), circulation Gamma, and the
 *     Vatistas core radius v_core (= 1.3*dt*Uinf, LUDVM.py:259-260); v_core = 0 means point
 *     vortices (the reference's `viscous != True` branch, LUDVM.py:562-563), for which a
 *     coincident source/target pair yields NaN exactly as the reference's 0/0 does.
 */
#ifndef LUDVM_HIP_H
#define LUDVM_HIP_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LUDVM_ABI_VERSION 5

enum {
  LUDVM_OK = 0,
  LUDVM_E_ARG = 1,      /* bad argument (null pointer, range outside the wake, ...) */
  LUDVM_E_HIP = 2,      /* a HIP runtime call failed; message has the HIP error string */
  LUDVM_E_NOMEM = 3,    /* device or host allocation failed */
  LUDVM_E_NODEVICE = 4, /* no usable gfx950 device */
  LUDVM_E_STATE = 5,    /* call not valid in the current state (e.g. wake not reserved) */
  LUDVM_E_COMM = 6      /* RCCL could not be opened, or one of its calls failed; message has its error string */
};

/* Arithmetic the pair sum is evaluated in. */
enum {
  LUDVM_PREC_F32 = 0,   /* fp32 arithmetic (headline kernel).  Wherever the library lays the positions out itself (host
                           float64 inputs, the resident wake) they are stored as fp32 offsets from the origin of their
                           256-vortex block and the origin difference is added once per block pair ("local origins"):
                           same speed as plain fp32, and a wake at |x| ~ 50 with vortices 1e-3 apart keeps ~1e-5 of
                           max|u| instead of ~1e-3.  Device fp32 inputs are used as they are. */
  LUDVM_PREC_F32X2 = 1, /* positions as hi+lo fp32 pairs: dx = (xh_p - xh_w) + (xl_p - xl_w), rest fp32.
                           Removes the cancellation error of |x| ~ 50 vs spacing ~ 1e-3. */
  LUDVM_PREC_F64 = 2    /* fp64 throughout (parity / debug mode, and the small chord-target calls) */
};

typedef struct ludvm_ctx ludvm_ctx;

/* ---- lifecycle -------------------------------------------------------------------------- */

int ludvm_abi_version(void);
/* Create a context bound to HIP device `device_ordinal` (must be a gfx950 part).  Owns one stream and its workspaces.
 * The library's results never depend on the environment: a production build reads LUDVM_RCCL_LIB (which librccl to open)
 * and LUDVM_COMM_FORCE (tests) and nothing else; the A/B switches of the measurement build (libludvm_hip_exp.so,
 * -DLUDVM_EXPERIMENTS) are listed in INTEGRATION.md. */
int ludvm_create(int device_ordinal, ludvm_ctx** out);
int ludvm_destroy(ludvm_ctx* ctx);
const char* ludvm_last_error(const ludvm_ctx* ctx);
/* Device facts for the roofline: CU count, max shader clock (kHz), HBM bytes, name (NUL-terminated). */
int ludvm_device_info(ludvm_ctx* ctx, int* cu_count, int* clock_khz, long long* hbm_bytes, char* name,
                      int name_len);
/* external != 0: use the caller's hipStream_t (e.g. torch.cuda.current_stream().cuda_stream) for all
 * subsequent launches -- a NULL handle then means the device's default (null) stream, which is what
 * torch's default stream is.  external == 0: go back to the context's own stream. */
int ludvm_set_stream(ludvm_ctx* ctx, void* hip_stream, int external);
int ludvm_synchronize(ludvm_ctx* ctx);
/* Launch shape knobs (0 keeps the built-in heuristic): targets per lane {1,2,4}, source splits.  For the symmetric
 * kernels the second number is the count of d-chunks (work items) per tile -- the rule gives at most 64 -- or, in the quad
 * variant, per quad of tiles, where k > 0 asks for k uniform chunks instead of the rule's tapered ones (long items
 * first, the last eighth of the ring offsets in short ones).  Results change only through the partition into fp32
 * partial sums. */
int ludvm_set_tuning(ludvm_ctx* ctx, int targets_per_lane, int source_splits);

/* Self-interaction launches (targets are exactly the sources: wake roll-up, all-pairs calls on one
 * array) may use the symmetric kernel, which evaluates each unordered pair once (K(i->j) = -K(j->i)): ~1.5x
 * faster.  It accumulates in 64-bit fixed point (integer atomics, scale from sum|Gamma| / v_core), so its results
 * do not depend on the order of the atomics: a launch repeats bit for bit, like the direct kernel and like the
 * reference.  Launches with v_core = 0 (no bound on the kernel) always take the direct kernel.
 * mode 0 = never (direct kernel), 1 = automatic (default; fp32, N >= 16384),
 * mode >= 2 = automatic with that value as the smallest N that takes the symmetric kernel. */
int ludvm_set_symmetric(ludvm_ctx* ctx, int mode);
/* Tuning of the symmetric kernel (tests; 0 = the library's rule by launch size): vortices per lane (4: 256-vortex tiles,
 * 8: 512-vortex tiles; plain fp32 positions only) and the number of wavefronts (1, 2, 4) that share the 64 rotation steps
 * of one tile pair.  Results change only through the partition into fp32 partial sums.  (The measurement build also takes
 * three negative codes that force a variant at every size; a production build answers them with LUDVM_E_ARG.) */
int ludvm_set_sym_tuning(ludvm_ctx* ctx, int vortices_per_lane, int rotation_split);

/* Sharding ONE simulation's roll-up over several GPUs (LUDVM.time_loop, LUDVM.py:1095-1127, with a wake too large
 * for one GPU to be quick about).  Every GPU holds the whole wake and runs the whole time loop -- chord sums, solve,
 * Euler step: everything that is O(N) or smaller -- but evaluates only tile block `rank` of `world` of the symmetric
 * kernel's unordered pairs, accumulating into d_acc (caller-owned device memory, e.g. a torch int64 tensor; at
 * least 16 (wake capacity + 64) + 16 bytes).  Before each Euler finisher the library calls
 *     allreduce(user, d_buf, count, hip_stream)
 * which must enqueue, on hip_stream, an in-place SUM all-reduce of the `count` 64-bit integers at d_buf (a prefix of
 * d_acc) over all owners, and return 0 (e.g. torch.distributed.all_reduce on RCCL).  Integer sums commute, so every
 * owner reads the same bits -- those one GPU owning all tiles would have produced -- and the replicated state cannot
 * drift apart.  Roll-ups of fewer than min_vortices vortices (a collective per step costs more than it saves there;
 * ~1e5 on xGMI) are done whole by every owner, without the hook.  world == 1 restores the unsharded behaviour.  On a
 * sharded context every symmetric launch of min_vortices or more is collective: all owners must issue the same calls
 * in the same order. */
typedef int (*ludvm_allreduce_fn)(void* user, void* d_buf, size_t count, void* hip_stream);
int ludvm_set_shard(ludvm_ctx* ctx, int rank, int world, size_t min_vortices, ludvm_allreduce_fn allreduce, void* user,
                    void* d_acc, size_t acc_bytes);

/* ---- the library's own communicator: one simulation, or one synthetic wake, on the GPUs of a node ----------------
 *
 * One process per GPU, each with its own context; the collectives run on RCCL over xGMI INSIDE the library, on the
 * context's stream -- a caller needs no communication library of its own, only a way to hand 128 bytes from rank 0 to
 * the other processes (a file, an environment variable, MPI, torch.distributed ...).  librccl.so is opened when the
 * first of these calls is made (environment LUDVM_RCCL_LIB names the file, default librccl.so.1); a process that never
 * calls them does not need RCCL.
 *
 *   ludvm_comm_unique_id   rank 0: the identifier of a new communicator (LUDVM_COMM_ID_BYTES bytes).
 *   ludvm_comm_init        every rank, with the same identifier: joins the communicator (returns when all `world` ranks
 *                          have) and shards the context's symmetric roll-ups over it exactly as ludvm_set_shard does
 *                          (LUDVM.time_loop's roll-up, LUDVM.py:1095-1127: every rank holds the whole wake and runs the
 *                          whole loop, but evaluates only tile block `rank` of `world` of the unordered pairs) -- with
 *                          the ONE collective per time step, an in-place ncclAllReduce (int64 sum) of the fixed-point
 *                          accumulators and their NaN counter, issued by the library between the symmetric kernel and the
 *                          Euler finisher.  Integer sums commute: every rank reads the bits one GPU would have produced.
 *                          Roll-ups of fewer than min_vortices vortices are done whole by every rank, without a collective.
 *   ludvm_comm_init_all    ONE PROCESS, several devices ("one RCCL communicator, single process, ncclCommInitAll"):
 *                          the n contexts -- one per device, ctxs[k] becomes rank k -- join a new communicator in one call
 *                          from one thread; no identifier, nothing to exchange.  Each context is sharded exactly as by
 *                          ludvm_comm_init.  From then on every context is driven by a HOST THREAD OF ITS OWN, which issues
 *                          the same calls in the same order as a process-per-GPU rank would (RCCL: one thread per device, or
 *                          grouped calls; the entry points that contain the collective -- ludvm_wake_advect*, ludvm_wake_step,
 *                          ludvm_march_run: LUDVM.time_loop's roll-up, LUDVM.py:1095-1127 -- enqueue it between their own
 *                          kernels, so grouping them from one thread would mean cutting each entry point in two).
 *                          ludvm_amd/multi.py is that driver: LUDVM(..., devices=[0, 1, ...]) with no launcher.  ABI 5.
 *   ludvm_comm_destroy     leaves the communicator (collective in RCCL's sense: every rank calls it) and unshards the context.
 *   ludvm_comm_info        rank and world of the context's communicator (world = 0: none).
 *   ludvm_comm_allreduce_i64_dev / ludvm_comm_allgather_dev
 *                          the two collectives of the synthetic-wake step (BASELINE config 4, ludvm_amd/sharded.py) on
 *                          caller-owned device buffers, asynchronous on the context's stream: in-place int64 sum (the
 *                          symmetric variant's accumulators) and an all-gather of bytes_per_rank bytes per rank into
 *                          d_recv[world * bytes_per_rank] (the direct variant's positions -- the north star's "single RCCL
 *                          all-gather of sources per step").
 *   ludvm_comm_allgather_host
 *                          the same all-gather for host buffers (synchronous): how the ranks of one simulation exchange
 *                          their blocks of a flow field (LUDVM.py:1186-1298) or of an induced_velocity call (:549-570).
 */
#define LUDVM_COMM_ID_BYTES 128
int ludvm_comm_unique_id(void* id_out, size_t id_bytes);
int ludvm_comm_init(ludvm_ctx* ctx, int rank, int world, const void* id, size_t id_bytes, size_t min_vortices);
int ludvm_comm_init_all(ludvm_ctx** ctxs, int n, size_t min_vortices);
int ludvm_comm_destroy(ludvm_ctx* ctx);
int ludvm_comm_info(ludvm_ctx* ctx, int* rank, int* world);
int ludvm_comm_allreduce_i64_dev(ludvm_ctx* ctx, long long* d_buf, size_t count);
int ludvm_comm_allgather_dev(ludvm_ctx* ctx, const void* d_send, void* d_recv, size_t bytes_per_rank);
int ludvm_comm_allgather_host(ludvm_ctx* ctx, const void* send, void* recv, size_t bytes_per_rank);

/* ---- stateless pair sum: backs LUDVM.induced_velocity (LUDVM.py:549-570) ------------------ */

/* u[p] =  sum_w g[w]*(zt[p]-zs[w]) / (2 pi sqrt(r^4 + vcore^4))
 * w[p] = -sum_w g[w]*(xt[p]-xs[w]) / (2 pi sqrt(r^4 + vcore^4)),  r^2 = dx^2+dz^2.
 * Host float64 in/out (what the reference passes and returns); `precision` selects the device
 * arithmetic.  ns == 0 or nt == 0 is valid (u, w zero-filled / untouched).
 * The reference's float64 sum (LUDVM.py:565-569) does not depend on the order of its arrays; LUDVM_PREC_F32 here keeps
 * 1e-5 of max|u| for ANY order: sources and targets that are not stored compactly (a user's array, a turbulence cloud of
 * :98-130 -- unlike a shed wake) are evaluated in Morton order on the device and the results returned in the caller's
 * order (the given order is kept, and with it every result bit, whenever it is already compact: within 3 x of an
 * area-filling arrangement, or of a line across its bounding box, or within 1.5 x of what Morton order achieves); a call
 * with fewer than 2048 sources or targets -- too few to make 128-point origin classes compact -- runs in float64 while
 * ns * nt <= 2^28 (~0.2 ms at most) and on hi+lo positions beyond (a few probe points in a wake of millions: 1.3 x the
 * fp32 time instead of the float64 rate on every pair); and a set too sparse for its core (mean class extent > 150 v_core
 * in Morton order, > 300 v_core for a set that is compact as given: fp32 offsets cannot resolve a core that small) takes
 * hi+lo positions as LUDVM_PREC_F32X2 does.  The arithmetic of a LUDVM_PREC_F32 call, and with it the last bits of its
 * result, therefore depends on the SIZES and the COMPACTNESS of what is passed (never on anything else: same arrays, same
 * bits); a set that evolves slowly across one of these thresholds changes route from one call to the next, within the
 * stated 1e-5 either way. */
int ludvm_induce_f64(ludvm_ctx* ctx, const double* xs, const double* zs, const double* gs, size_t ns,
                     const double* xt, const double* zt, size_t nt, double vcore, int precision,
                     double* u, double* w);
/* The order the library itself would evaluate the n points (x, z) in: order[k] = index of the point that takes position
 * k.  Morton order when the given order is not compact (see ludvm_induce_f64), else the identity (*reordered = 0; also
 * for n < 2048).  For callers of the RESIDENT wake, whose slots are the caller's indices: LUDVM uploads a cloud of free
 * vortices (LUDVM.py:98-130, :274-277) in this order and returns history rows in its own, so that fp32 roll-ups of an
 * unordered cloud keep the accuracy tier of a shed wake.  *mean_class_extent (may be NULL) = mean over the 128-point origin
 * classes, in that order, of (xmax - xmin) + (zmax - zmin); fp32 on local origins keeps 1e-5 of max|u| up to about
 * 150 v_core for a cloud that had to be reordered, 300 v_core for a set that is compact as given (beyond:
 * LUDVM_PREC_F32X2).  Deterministic. */
int ludvm_spatial_order(ludvm_ctx* ctx, const double* x, const double* z, size_t n, unsigned* order, int* reordered,
                        double* mean_class_extent);
/* Same with host float32 buffers (always LUDVM_PREC_F32 arithmetic). */
int ludvm_induce_f32(ludvm_ctx* ctx, const float* xs, const float* zs, const float* gs, size_t ns,
                     const float* xt, const float* zt, size_t nt, float vcore, float* u, float* w);
/* Device-resident fp32 SoA (torch tensors): asynchronous on the context stream.  Used by the
 * benchmark, the multi-GPU shard step and the flow-field path. */
int ludvm_induce_dev_f32(ludvm_ctx* ctx, const float* d_xs, const float* d_zs, const float* d_gs, size_t ns,
                         const float* d_xt, const float* d_zt, size_t nt, float vcore, float* d_u,
                         float* d_w);
/* One explicit-Euler advection step on device fp32 SoA (LUDVM.py:1105-1127 with the wake as both
 * source and target set): (u,w) induced by all ns sources on targets [t_first, t_first+nt) of the
 * same arrays, then x_out[i] = x[t_first+i] + dt*u[i], z_out likewise.  x_out/z_out must not alias
 * the source arrays (the multi-GPU step all-gathers them into the next source buffer). */
int ludvm_advect_dev_f32(ludvm_ctx* ctx, const float* d_xs, const float* d_zs, const float* d_gs, size_t ns,
                         size_t t_first, size_t nt, float vcore, float dt, float* d_x_out, float* d_z_out);

/* Multi-GPU building blocks of the symmetric roll-up (ludvm_amd/sharded.py: BASELINE config 4, bench.py's N > 1 line).
 * Reading the benchmark across GPU counts: bench.py's `value` at N = 1 is config 3 (N = 1e6, one ludvm_induce_dev_f32 call per
 * step); at N > 1 it is config 4 (N = 8e6, these entry points + one collective per step).  The same-work denominator of
 * "8 GPUs vs 1" is the N = 1 line's `config4_one_gpu.value` (config 4's step on one GPU), which every N > 1 line names in
 * `scaling_denominator` -- not the N = 1 line's `value`.
 * Vortices are cut into
 * tiles of LUDVM_SYM_TILE; the caller owns tiles [tile_first, tile_first + tile_count) of the
 * ceil(n / LUDVM_SYM_TILE) tiles.
 *   ludvm_sym_scale_dev_f32 derives the fixed-point scale of the raw sums from sum|Gamma| / v_core (summed in a
 *     fixed order: every GPU holding the same circulations gets the same record) into d_scale, a caller-owned
 *     device record of LUDVM_SYM_SCALE_BYTES; v_core must be > 0.
 *   ludvm_sym_accumulate_dev_f32 evaluates this owner's share of the unordered pairs (its tiles against the cyclic
 *     half of the tile ring) and ADDS the raw sums of both partners, as 64-bit fixed-point integers, into
 *     d_acc_u / d_acc_w (n each, zeroed by the caller); *d_bad (zeroed by the caller) is incremented when a partial
 *     sum was not finite.  Integer sums are order-independent: summed over all owners (a sum all-reduce of the
 *     three buffers) they are bit for bit what one GPU owning all tiles produces.
 *   ludvm_advect_from_sums_dev_f32 turns the summed values of targets [t_first, t_first + nt) (d_sum_*[i] belongs to
 *     target t_first + i) into the Euler step x_out[i] = x[t_first + i] + dt * u, z_out likewise
 *     (LUDVM.py:1108-1109); NaN when *d_bad != 0, as the reference's sum over a NaN source would be. */
#define LUDVM_SYM_TILE 512
#define LUDVM_SYM_OWNER_ALIGN 4   /* an owner's tile block starts and ends on multiples of this many tiles (or ends with the
                                     ring): large launches add the partial sums of four consecutive tiles' wavefronts in fp32
                                     before they are converted to fixed point, so such a quad must not be cut between owners */
#define LUDVM_SYM_SCALE_BYTES 32
int ludvm_sym_scale_dev_f32(ludvm_ctx* ctx, const float* d_g, size_t n, float vcore, void* d_scale);
int ludvm_sym_accumulate_dev_f32(ludvm_ctx* ctx, const float* d_x, const float* d_z, const float* d_g, size_t n,
                                 size_t tile_first, size_t tile_count, float vcore, const void* d_scale,
                                 long long* d_acc_u, long long* d_acc_w, long long* d_bad);
int ludvm_advect_from_sums_dev_f32(ludvm_ctx* ctx, const long long* d_sum_u, const long long* d_sum_w,
                                   const void* d_scale, const long long* d_bad, const float* d_x, const float* d_z,
                                   size_t t_first, size_t nt, float dt, float* d_x_out, float* d_z_out);

/* ---- resident wake: backs LUDVM.time_loop (LUDVM.py:597-1171) ----------------------------- */
/* The wake (TEV, LEV and FREE vortices, in an order the host chooses) lives on the device across
 * time steps as float64 master copies plus the fp32 (hi, lo) SoA the pair kernel reads. */

int ludvm_wake_reserve(ludvm_ctx* ctx, size_t capacity);      /* grow-only; keeps contents */
int ludvm_wake_clear(ludvm_ctx* ctx);                          /* size := 0 */
int ludvm_wake_size(ludvm_ctx* ctx, size_t* n);
/* Drop the vortices past the first n (n <= size).  The phantom LEV slot of a non-shedding step
 * (LUDVM.py:1112-1118 moves slot ilev although nothing was shed) is appended, advected, read back
 * and dropped this way. */
int ludvm_wake_truncate(ludvm_ctx* ctx, size_t n);
/* Append `count` vortices (new TEV / LEV placement, LUDVM.py:672-681, 788-800). */
int ludvm_wake_append(ludvm_ctx* ctx, const double* x, const double* z, const double* gamma, size_t count);
/* Overwrite positions and/or circulation of [first, first+count); NULL leaves that field as is.
 * The circulation of the newest TEV/LEV is solved after placement (LUDVM.py:758-760, 953-954). */
int ludvm_wake_write(ludvm_ctx* ctx, size_t first, size_t count, const double* x, const double* z,
                     const double* gamma);
/* Read back [first, first+count) (history rows of path['TEV'|'LEV'|'FREE'], LUDVM.py:615-618).
 * Any of x, z, gamma may be NULL. */
int ludvm_wake_read(ludvm_ctx* ctx, size_t first, size_t count, double* x, double* z, double* gamma);
/* Velocity induced by wake vortices [src_first, src_first+src_count) at nt host points (the
 * Npoints-1 bound-vortex points of the chord: LUDVM.py:582-584, 1049-1054).  fp64 arithmetic:
 * these O(Npanels x Nw) sums feed the Gamma solve and the loads and cost <1 % of a step. */
int ludvm_wake_induce_on_points(ludvm_ctx* ctx, size_t src_first, size_t src_count, const double* xt,
                                const double* zt, size_t nt, double vcore, double* u, double* w);
/* One call for everything a time step needs at the chord before the Gamma solve: the sum of
 * ludvm_wake_induce_on_points over wake vortices [src_first, src_first+src_count) -> (u_wake, w_wake)[nt],
 * plus the velocity induced there by n_unit (<= 4) unit-strength vortices at (unit_x, unit_z) -- the new
 * TEV and the candidate LEV (LUDVM.py:751, :926, :931) -> u_unit, w_unit as [n_unit][nt] rows.  fp64.
 * One host->device copy, one device->host copy, one synchronisation. */
int ludvm_wake_chord_sums(ludvm_ctx* ctx, size_t src_first, size_t src_count, const double* xt, const double* zt,
                          size_t nt, const double* unit_x, const double* unit_z, size_t n_unit, double vcore,
                          double* u_wake, double* w_wake, double* u_unit, double* w_unit);
/* Wake roll-up, fused (LUDVM.py:1095-1127): for every wake vortex i in [0, size):
 *   (u,w)_i = induced by all wake vortices  +  induced by the nfoil bound vortices (foil_x, foil_z,
 *   foil_dgamma: LUDVM.py:1096,1099-1100), both with the same core radius;
 *   x_i += dt*u_i ; z_i += dt*w_i   (explicit Euler, float64 update of the master copy).
 * `precision` selects the pair arithmetic.  If u_out/w_out are non-NULL the induced velocities
 * (before the update) are also returned (size doubles each). */
int ludvm_wake_advect(ludvm_ctx* ctx, double dt, const double* foil_x, const double* foil_z,
                      const double* foil_dgamma, size_t nfoil, double vcore, int precision, double* u_out,
                      double* w_out);
/* Same, and read back the updated positions of the last tail_count wake vortices (the newest TEV / LEV,
 * which place the next ones: LUDVM.py:680-681, 797-798) in the same call; synchronous when
 * tail_count > 0. */
int ludvm_wake_advect_tail(ludvm_ctx* ctx, double dt, const double* foil_x, const double* foil_z,
                           const double* foil_dgamma, size_t nfoil, double vcore, int precision, size_t tail_count,
                           double* tail_x, double* tail_z);

/* One round trip per time step -- one packed upload, one staging kernel, the pair kernels, one download:
 *   1. append the n_new vortices shed this step (new_x/z/gamma; LUDVM.py:672-681, 788-800, 953-954);
 *   2. the roll-up of step i exactly as ludvm_wake_advect (LUDVM.py:1095-1127);
 *   3. what step i+1 needs before its Gamma solve:
 *      - placement of the next TEV and of the candidate LEV from the advected positions: one third of the
 *        way from the trailing edge te[2] / leading edge le[2] (of step i+1) to the newest TEV (vortex
 *        size - tail_count) / newest LEV (vortex size - 1, used when lev_from_prev != 0 and tail_count == 2;
 *        otherwise the candidate sits on the leading edge)  (LUDVM.py:680-681, :788-800);
 *      - ludvm_wake_chord_sums over the whole advected wake at the nt points (xt, zt) of step i+1 with those
 *        two unit vortices.
 * Outputs: tail_x/z[tail_count] (updated newest vortices), unit_x/z[2] (the placed TEV, LEV candidate),
 * u_wake/w_wake[nt], u_unit/w_unit[2][nt].  tail_count is 1 or 2. */
int ludvm_wake_step(ludvm_ctx* ctx, const double* new_x, const double* new_z, const double* new_gamma, size_t n_new,
                    double dt, const double* foil_x, const double* foil_z, const double* foil_dgamma, size_t nfoil,
                    double vcore, int precision, const double* te, const double* le, int lev_from_prev,
                    size_t tail_count, const double* xt, const double* zt, size_t nt, double* tail_x, double* tail_z,
                    double* unit_x, double* unit_z, double* u_wake, double* w_wake, double* u_unit, double* w_unit);

/* ---- device-resident time march: many steps of LUDVM.time_loop per call (LUDVM.py:597-1171) --------
 *
 * ludvm_wake_step still costs one host round trip per time step, because the Gamma solve between two
 * roll-ups (LUDVM.py:743-1090) runs on the host.  These two calls move that solve to the device: 'Faure' method,
 * closed-form Gamma_TEV :758-760 and the 2x2 TEV/LEV system :944-954; 'Ramesh', the Newton iterations of :683-739
 * and :807-914 on the six projections of T1, T2, T3 (the downwash is linear in the circulations); for both, the
 * Fourier coefficients :765-773, LESP criterion :781-805, bound vorticity :987-1010 and loads :1035-1090.  So `count`
 * consecutive steps are enqueued back to back; whether a LEV is shed -- and hence the wake size -- is decided on the
 * device.
 *
 * ludvm_march_setup uploads what does not change during a run:
 *   scalars[12] = Uinf, chord, rho, dt, piv, v_core, IC (:647), sum(Gamma_free), method (0 'Faure', 1 'Ramesh'),
 *                 maxerror, maxiter, epsilon (the Newton controls of 'Ramesh', :253-255; ignored for 'Faure');
 *   tables     = detadx_panel[np] | eta_panel[np] | x_panel[np] | cm1[np] | wq[np] | opcs[np] | hcsd[np] | wx[np]
 *                | cproj[nc][np] | ssin[nc-1][np]
 *                with np = npan, nc = ncoef and, on theta_panel with trapezoid weights wq:
 *                cm1 = (cos(theta) - 1) wq, opcs = (1 + cos)/sin, hcsd = c/2 sin(theta) dtheta, wx = trapezoid
 *                weights on x_panel, cproj[0] = -wq/pi, cproj[n] = 2/pi cos(n theta) wq, ssin[n-1] = sin(n theta);
 *   kin        = one row per time step i, [alpha, alpha_dot, h_dot, te_x, te_z, le_x, le_z, xg[np], zg[np]]
 *                (airfoil_gamma_points and the edges of path['airfoil'] at step i).
 * 1 <= npan <= 256, 4 <= ncoef <= 64.
 *
 * ludvm_march_run advances the resident wake through time steps [first_step, first_step + count), all inside
 * the kinematics table.  state (16 + ncoef doubles, in and out):
 *   [0] wake size (in: must equal ludvm_wake_size)   [1] TEVs shed   [2] LEVs shed
 *   [3] a LEV was shed in the previous step          [4] LESPcrit with its current sign (:802-805)
 *   [5] sum Gamma_TEV   [6] sum Gamma_LEV            [7..10] tev_x, lev_x, tev_z, lev_z of the coming step
 *   [11] out: vortices shed by the last step (1 or 2)
 *   [12..15] out: x[size-2], x[size-1], z[size-2], z[size-1] after the last roll-up (ignored on input)
 *   [16..16+ncoef) Fourier coefficients of the previous step.
 * rows (out): count rows of 12 + 2 ncoef + 2 npan doubles:
 *   g_tev, g_lev, shed(0/1), bound, LESP_prev, LESP, Fn, Fs, M, wake slot of the new TEV,
 *   velocity (u, w) at the origin on a step that sheds no LEV (the reference convects a zero-strength LEV slot
 *   from there and stores where it lands, :1112-1118), A[ncoef], dA/dt[ncoef], gamma[npan], dGamma[npan].
 * hist (out, may be NULL): the reference's dense trajectory history (:1108-1127) -- after each step's roll-up the
 *   positions of all wake vortices, count rows of x[hist_nmax] | z[hist_nmax] in wake (shedding) order;
 *   hist_nmax >= wake size + 2 count.
 * anchors (in, may be NULL): the wake sizes after the four anchor steps max(64 (floor(first_step / 64) - 3 + q) - 1, 0),
 *   q = 0 .. 3 (the caller knows the wake size after every step it has run; step 0 = the initial wake); -1 = not given.
 *   Each step's launch geometry is derived from the wake size two 64-step periods back -- read from the device inside a
 *   call, given here across calls -- so a run's bits do not depend on where its calls begin.  Every given size is checked
 *   against state[0] (one or two vortices per step since the anchor), LUDVM_E_ARG otherwise.  NULL / all -1: the bounds
 *   restart at this call (results then depend on the chunking, to fp32 rounding, near the tile thresholds).
 * Synchronous: returns when the last step has finished.  `precision` selects the roll-up arithmetic as in
 * ludvm_wake_advect; the solve is float64.  From the symmetric-kernel threshold on, a step's chord sums and solve run on
 * a second stream beside the symmetric kernel.  The launch
 * geometry of every step is a function of the step number and of the simulation itself (never of host timing, nor -- with
 * `anchors` given -- of where the calls begin), and every sum is order-independent or done in a fixed order: two runs
 * return the same bits, however they are cut into calls. */
int ludvm_march_setup(ludvm_ctx* ctx, int npan, int ncoef, const double* scalars, const double* tables, const double* kin,
                      size_t kin_rows);
int ludvm_march_run(ludvm_ctx* ctx, long long first_step, long long count, int precision, double* state, double* rows,
                    double* hist, size_t hist_nmax, const long long* anchors);

/* ---- flow field: backs LUDVM.flowfield (LUDVM.py:1186-1298) -------------------------------- */

/* Grid targets generated on the device, x-major ravel like np.meshgrid(indexing='ij')
 * (LUDVM.py:1193-1195): point (i,j) = (xmin + i*dr, zmin + j*dr), i < nx, j < nz, index i*nz + j.
 * Sources are host float64 arrays (wake ++ foil as the caller gathered them, LUDVM.py:1202-1217).
 * Outputs u, w are host float32 arrays of nx*nz (fp32 arithmetic on local-origin positions: the sources as offsets
 * from the origin of their 256-source blocks, the grid points generated in float64 and referred to those origins).
 * d_u/d_w variants keep the result on the device for the vorticity stencil. */
int ludvm_flowfield_f32(ludvm_ctx* ctx, double xmin, double zmin, double dr, size_t nx, size_t nz,
                        const double* xs, const double* zs, const double* gs, size_t ns, double vcore,
                        float* u, float* w);
/* Velocity field and its vorticity (LUDVM.py:1224-1292) in one call: the stencil runs on the device on the
 * fields where they are, and u, w, ome (host float32, nx*nz each; ome may be NULL) come back together. */
int ludvm_flowfield_vorticity_f32(ludvm_ctx* ctx, double xmin, double zmin, double dr, size_t nx, size_t nz,
                                  const double* xs, const double* zs, const double* gs, size_t ns, double vcore,
                                  float* u, float* w, float* ome);
/* Rows [row_first, row_first + row_count) of that nx x nz grid only -- the same numbers, bit for bit, that the
 * whole-grid call returns for them (the grid coordinates are generated from the global row index and the split of the
 * sources into partial sums is planned from the whole grid, not from the block; with ome wanted, one halo row per
 * interior side is evaluated internally).  Outputs are row_count * nz each.  This is the unit a
 * multi-GPU flow field shards by: each GPU evaluates a block of rows, nothing is exchanged. */
int ludvm_flowfield_rows_f32(ludvm_ctx* ctx, double xmin, double zmin, double dr, size_t nx, size_t nz, size_t row_first,
                             size_t row_count, const double* xs, const double* zs, const double* gs, size_t ns,
                             double vcore, float* u, float* w, float* ome);
/* The same rows in float64 throughout -- pair sums, grid coordinates and the vorticity stencil with the mesh differences
 * taken from the mesh values, as the reference evaluates flowfield (LUDVM.py:1206, :1216-1217, :1224-1292): u, w and ome
 * then equal the reference's to rounding (~1e-13 of their maxima).  Outputs are host float64 arrays of row_count * nz;
 * the whole grid is row_first = 0, row_count = nx.  LUDVM.flowfield uses it when the run's precision is 'f64'. */
int ludvm_flowfield_rows_f64(ludvm_ctx* ctx, double xmin, double zmin, double dr, size_t nx, size_t nz, size_t row_first,
                             size_t row_count, const double* xs, const double* zs, const double* gs, size_t ns,
                             double vcore, double* u, double* w, double* ome);
/* Same as ludvm_flowfield_f32, device-resident fp32 sources and outputs (asynchronous). */
int ludvm_flowfield_dev_f32(ludvm_ctx* ctx, double xmin, double zmin, double dr, size_t nx, size_t nz,
                            const float* d_xs, const float* d_zs, const float* d_gs, size_t ns, float vcore,
                            float* d_u, float* d_w);
/* Vorticity dw/dx - du/dz on the uniform grid: centred differences inside, one-sided on edges and
 * corners (LUDVM.py:1224-1292).  Host float32 in/out, nx*nz each. */
int ludvm_vorticity_f32(ludvm_ctx* ctx, const float* u, const float* w, size_t nx, size_t nz, double dr,
                        float* ome);
int ludvm_vorticity_dev_f32(ludvm_ctx* ctx, const float* d_u, const float* d_w, size_t nx, size_t nz, float dr,
                            float* d_ome);

/* ---- measurement ---------------------------------------------------------------------------- */

/* Probe of the symmetric kernel's fixed-point accumulation (the order-independent sums behind the roll-up of
 * LUDVM.py:1095-1127): units[i] = the 64-bit integer the kernel adds to an accumulator for the fp32 partial sum
 * values[i] at scale 2^scale_log2, i.e. trunc(values[i] * 2^scale_log2) -- exact for every |values[i] * 2^scale_log2|
 * < 2^63.  Host arrays; the conversion runs on the device with the kernel's own code. */
int ludvm_fixed_point_probe(ludvm_ctx* ctx, const float* values, size_t n, int scale_log2, long long* units);

/* Average device time (ms) of the pair kernel launches (main kernel only -- direct or symmetric --
 * not the split reduction / finisher) issued since the last call with reset != 0, measured with HIP events on the stream the
 * kernel is launched on; *launches = number of launches averaged.  Timing is off until
 * ludvm_kernel_timing(ctx, 1). */
int ludvm_kernel_timing(ludvm_ctx* ctx, int enable);
int ludvm_kernel_time_ms(ludvm_ctx* ctx, int reset, double* avg_ms, long long* launches);

#ifdef __cplusplus
}
#endif
#endif /* LUDVM_HIP_H */
