/* include/lsq_hip.h -- the drop-in boundary of the MI355X LSQ fake-quantize hot path.
 *
 * A plain C ABI (extern "C", raw device pointers and sizes, no torch types) exported by
 * `liblsq_hip.so` (built from lsqfakequantize-pytorch_amd/csrc/ for gfx950).  These entry points
 * are exactly what a binding for the reference's backend kernels would call: each one replaces
 * one `TORCH_LIBRARY_IMPL(torchlsq, CUDA)` kernel of the reference
 *   /root/reference/torchlsq/csrc/ops/cuda/lsq_cuda.cu:301-314
 * (whose CPU twins, lsq_cpu.cpp:298-311, are the parity oracle) -- see INTEGRATION.md for the
 * binding a maintainer of the reference would add.
 *
 * Contract:
 *  - the caller owns every buffer (outputs and workspace included); the library never allocates
 *    or frees device memory and keeps no reference after return;
 *  - all pointers are DEVICE pointers on the current HIP device; kernels are enqueued on the
 *    `stream` argument (a hipStream_t passed as void*; NULL = the default stream) and the call
 *    returns without synchronising -- scale/shift are read on the device, never on the host
 *    (the reference's `scale[0].item()` host syncs, lsq_cuda.cu:52-53,120-121, are gone).
 *    The exceptions are the SETUP calls of the rank communicator, each documented where it is declared:
 *    lsq_hip_comm_create / _destroy (RCCL's own bootstrap and teardown), lsq_hip_comm_tune (scratch
 *    memory, synchronises) and lsq_hip_comm_configure (synchronises the communicator's stream).  No
 *    entry point on the per-step path -- the ops, lsq_hip_comm_all_reduce*, _join, _sharded_finish --
 *    allocates or synchronises;
 *  - no behaviour depends on the process environment: the library reads no environment variable
 *    (tests/test_abi.py checks that the shipped binary does not import getenv); what can be chosen is
 *    an argument or a field of an options struct;
 *  - re-entrant, no mutable global state: forward (caller thread) and backward (autograd engine
 *    thread) may run concurrently.  The library keeps only fill-once lookup tables (device
 *    properties, kernel register counts, per-thread workspace sizes per shape) and exports nothing
 *    but the symbols declared here -- no tuning or debug entry point (tests/test_abi.py);
 *  - functions return 0 on success, a negative LSQ_E* code for rejected arguments, or a positive
 *    hipError_t value; they never throw.  lsq_hip_last_error() describes the last failure of the
 *    calling thread.
 *
 * Memory layout: tensors are dense in memory order.  Per-tensor ops see a flat array of `n`
 * elements.  Per-channel ops see the 3-D view [outer, C, inner] of the same dense memory (the
 * view reference lsq_cpu.cpp:168-176 builds by reshaping scale/shift to [1,..,C,..,1]); e.g. an
 * NCHW activation quantised on axis 1 is (N, C, H*W), a conv weight on axis 0 is (1, Cout, Cin*kh*kw),
 * a channels-last activation is (N*H*W, C, 1).
 */
#ifndef LSQ_HIP_H_
#define LSQ_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LSQ_HIP_ABI_VERSION 6

/* Storage type of x / grad / y / dx.  Arithmetic is fp32 for F32, BF16 and F16 storage and fp64
 * for F64 (reference CPU path: AT_DISPATCH_FLOATING_TYPES, lsq_cpu.cpp:37,92,182,243).
 * scale / shift / ds / db are float for F32|BF16|F16 storage and double for F64. */
enum lsq_dtype {
    LSQ_F32 = 0,
    LSQ_F64 = 1,
    LSQ_BF16 = 2, /* extension (BASELINE config 5): bf16 in/out, fp32 math, RNE on store.  d_scale / d_shift of 16-bit storage:
                   * the window kernels on the LDS ring add the terms of up to 4 consecutive rows per packet component in fp32
                   * before they join the fp64 accumulator (and apply the gradient scaler to the sums, not to each term), so
                   * the result sits up to 1.8e-7 x sum|terms| (measured 6e-8) from the exact sum -- inside the 1e-6 parity
                   * budget, but on heavily cancelling sums (|sum| << sum|terms|) visibly further from it in RELATIVE terms
                   * than the fp32-storage kernels are (profiles/r03_reduction_margin.txt: 4.7e-4 vs 5.8e-8 on config 5's
                   * mixed-sign d_scale).  Parity for this storage type is defined by this build (no reference exists). */
    LSQ_F16 = 3   /* extension: fp16 in/out, fp32 math (the reference's CUDA path computes in half) */
};

enum lsq_status {
    LSQ_OK = 0,
    LSQ_EINVAL = -1,    /* bad argument (null pointer, negative size, unknown dtype, ...) */
    LSQ_EWORKSPACE = -2, /* workspace too small or misaligned */
    LSQ_ECOMM = -3       /* rank communicator: RCCL missing or an RCCL call failed (lsq_hip_last_error has its message) */
};

/* Scalar arguments shared by the four ops; field-for-field the trailing arguments of the
 * reference schemas (lsq.cpp:138-145).  `numel_for_scaler` is new: the element count used in the
 * gradient scaler 1/sqrt(numel*quant_max[/C]) (lsq_cpu.cpp:103,250).  <= 0 means "this call's own
 * numel" (the reference behaviour); the batch-sharded multi-GPU path passes the GLOBAL numel so
 * that the summed shards equal the unsharded result. */
typedef struct lsq_params {
    int32_t quant_min;
    int32_t quant_max;
    int32_t type_min;
    int32_t type_max;
    int32_t use_grad_scaling;
    int32_t sym;       /* 1: symmetric, d_shift forced to 0 (lsq_kernel.h:118) */
    int32_t eval_mode; /* 1: backward produces dx only, ds = db = 0 (lsq_kernel.h:126-145) */
    int32_t init_mode; /* 1: forward is the identity, backward uses 2*(x_r - x) (lsq_kernel.h:13,116) */
    double grad_scaler;
    int64_t numel_for_scaler;
} lsq_params;

/* Optional extra output of the forward: ONE auxiliary byte per element, laid out like y (NULL = not wanted).
 *  aux_kind 0 -- integer levels: q = rne(clamp(x/s + zp, quant_min, quant_max))  (lsq_kernel.h:13), stored
 *                as the byte (q - level_bias) mod 256: an int8 when [quant_min, quant_max] - level_bias lies in
 *                -128..127 (qint8 levels as they are; quint8 levels with level_bias 128), a uint8 when it lies in 0..255
 *                (quint8 levels as they are, level_bias 0: the int_repr of a torch.quint8 tensor).
 *  aux_kind 1 -- inside mask: 1 where quant_min < clamp(..) < quant_max strictly, else 0: everything the
 *                eval-mode backward needs (lsq_kernel.h:126-145), see lsq_hip_backward_from_mask.
 * With `levels` set the forward's `y` argument may be NULL: only the bytes are written (5 instead of 9 bytes of traffic per
 * fp32 element) -- the conversion-time pass that turns a trained fake-quantizer's input into a real quantized tensor
 * (reference quantized/modules/observers.py:378-422 supplies scale / zero_point for it) needs nothing else. */
typedef struct lsq_fwd_extras {
    void* levels;
    int32_t level_bias;
    int32_t aux_kind;
} lsq_fwd_extras;

/* Optional extra of the backward: a TICKET, LSQ_TICKET_BYTES of device memory owned by the caller that outlives the
 * call.  With a ticket the backward is ONE launch: the workgroup that finishes last (an agent-scope atomic counter in
 * the ticket tells it so) folds the per-workgroup partial sums -- in their fixed index order, so d_scale / d_shift stay
 * bit-reproducible -- and stores d_scale / d_shift itself; without one (extras or ticket NULL) a second, tiny launch
 * does that (2-4 us on a small tensor: half of the backward of a BASELINE-config-1-sized activation).
 * Contract: all zero before its first use; every launch leaves it all zero again (the counters wrap), so it is reused
 * as is; launches that may run CONCURRENTLY (different streams, or a captured graph replayed next to eager work) must not
 * share a ticket -- keep one per stream and give captured launches none.
 * Honoured by lsq_hip_backward_per_tensor only.  lsq_hip_backward_per_channel accepts the argument and IGNORES it (it always
 * takes the two-launch route -- or a single launch where one workgroup owns a whole channel, e.g. conv / linear weights):
 * measured on MI355X the folded finalize is not faster than the finalize launch -- for the per-tensor kernel (DESIGN.md
 * section 4; the host layers still use it for small tensors, where the saved launch is host time) and for a per-window fold
 * built for the window kernels (DESIGN_HISTORY.md section 7, profiles/r03_pc_fused_fold_ab.txt) -- so the per-channel kernels have none. */
#define LSQ_TICKET_BYTES 4096
typedef struct lsq_bwd_extras {
    void* ticket;
} lsq_bwd_extras;

/* ---- library / build information -------------------------------------------------------- */

/* LSQ_HIP_ABI_VERSION the library was built with. */
int lsq_hip_abi_version(void);

/* HIP runtime version the library was compiled against (HIP_VERSION), the counterpart of the
 * reference's torchlsq::_cuda_version (torchlsq.cpp:25-31, which returns CUDA_VERSION or -1). */
int64_t lsq_hip_runtime_version(void);

/* Message for the last non-zero status returned to the calling thread ("" if none). */
const char* lsq_hip_last_error(void);

/* The gradient scaler of lsq_cpu.cpp:103-104 (per_channel = 0) / :250-251 (per_channel = 1) with
 * the reference's precision chain for the given dtype's arithmetic type; returned widened to
 * double.  Host-only helper (no GPU needed). */
double lsq_hip_grad_scaler(int dtype, int per_channel, int64_t numel, int32_t quant_max,
                           int64_t channels, int32_t use_grad_scaling, double grad_scaler);

/* ---- host-side launch policy shared by every host layer ------------------------------------------------ */

/* Two host layers sit on this ABI (the Python / ctypes one and the C++ torch binding); the decisions they
 * must make IDENTICALLY live here, as pure functions (no GPU, no state), so that they cannot drift apart:
 *
 * lsq_hip_policy_ticket: should this backward be given a ticket (one launch instead of kernel + finalize launch)?
 *   mode 0 never, 1 always, 2 "auto": per-tensor tensors of at most 8 MB -- host-bound in eager mode, where one launch less is
 *   11-15 % of a forward + backward step, while on the GPU the one-launch route is never faster (DESIGN_HISTORY.md section 4).  The
 *   per-channel entry point ignores tickets, so auto never asks for one there.  Captured launches take none whatever the
 *   answer (the caller knows whether its stream is capturing).
 * lsq_hip_policy_saves_mask: does the forward of an autograd node save the one-byte inside mask (aux_kind 1) instead of x?
 *   Only the eval-mode backward can run from it (lsq_kernel.h:126-145), only when an input gradient will be asked for, and
 *   only if the caller did not ask for the reference's behaviour to the letter (mask from the parameters at BACKWARD time,
 *   lsq_autograd.cpp:46-73: what LSQFakeQuantizer needs while its observer rewrites them). */
int lsq_hip_policy_ticket(int32_t mode, int32_t per_channel, int64_t tensor_bytes);
int lsq_hip_policy_saves_mask(int32_t eval_mode, int32_t init_mode, int32_t input_requires_grad, int32_t mask_backward);

/* ---- per-tensor ------------------------------------------------------------------------- */

/* Bytes of scratch the backward needs (block partial sums); 256-byte aligned buffer expected.
 * Does not depend on the data, only on the size; never more than 1 MiB. */
size_t lsq_hip_backward_per_tensor_workspace(int dtype, int64_t n);

/* y = fake_quant(x).  Replaces lsq_forward_per_tensor_impl, lsq_cuda.cu:18-61 (CPU twin
 * lsq_cpu.cpp:15-53).  scale, shift: device pointers to 1 element. */
int lsq_hip_forward_per_tensor(int dtype, const void* x, void* y, int64_t n, const void* scale,
                               const void* shift, const lsq_params* p, const lsq_fwd_extras* extras,
                               void* stream);

/* dx, ds[1], db[1] from grad and x in ONE pass over HBM (+ a tiny finalize launch when no ticket is given).
 * Replaces lsq_backward_per_tensor_impl, lsq_cuda.cu:64-143 (three elementwise kernels, three
 * N-sized temporaries and two at::sum; CPU twin lsq_cpu.cpp:56-141).
 * dsdb_wide (optional, NULL ok): double[2] = {sum ds terms, sum db terms} before rounding to the
 * parameter type -- what the sharded path all-reduces. */
int lsq_hip_backward_per_tensor(int dtype, const void* grad, const void* x, void* dx, void* ds,
                                void* db, double* dsdb_wide, int64_t n, const void* scale,
                                const void* shift, const lsq_params* p, const lsq_bwd_extras* extras,
                                void* workspace, size_t workspace_bytes, void* stream);

/* ---- per-channel ------------------------------------------------------------------------ */

size_t lsq_hip_backward_per_channel_workspace(int dtype, int64_t outer, int64_t channels, int64_t inner);

/* Replaces lsq_forward_per_channel_impl, lsq_cuda.cu:147-199 (CPU twin lsq_cpu.cpp:145-193).
 * scale, shift: device pointers to `channels` elements. */
int lsq_hip_forward_per_channel(int dtype, const void* x, void* y, int64_t outer, int64_t channels,
                                int64_t inner, const void* scale, const void* shift,
                                const lsq_params* p, const lsq_fwd_extras* extras, void* stream);

/* Replaces lsq_backward_per_channel_impl, lsq_cuda.cu:202-297 (CPU twin lsq_cpu.cpp:197-294).
 * ds, db: `channels` elements each.  dsdb_wide (optional): double[2*channels], ds sums first.
 * NOTE the gradient scaler follows the CPU oracle (/C inside the sqrt, lsq_cpu.cpp:250), not the
 * reference's CUDA kernel (lsq_cuda.cu:274), which omits it. */
int lsq_hip_backward_per_channel(int dtype, const void* grad, const void* x, void* dx, void* ds,
                                 void* db, double* dsdb_wide, int64_t outer, int64_t channels,
                                 int64_t inner, const void* scale, const void* shift,
                                 const lsq_params* p, const lsq_bwd_extras* extras, void* workspace,
                                 size_t workspace_bytes, void* stream);

/* ---- batch-sharded backward: the epilogue after the all-reduce --------------------------------------- */

/* One process per GPU, the batch (dim 0) split over the ranks: forward and dx need no communication, d_scale / d_shift are
 * sums over ALL ranks' elements and their gradient scaler 1/sqrt(numel*quant_max[/C]) (lsq_cpu.cpp:103,250) wants the
 * GLOBAL element count.  When every rank knows that count up front it goes into lsq_params.numel_for_scaler and the
 * all-reduced dsdb_wide only has to be rounded.  When it does not (uneven shards, e.g. the last batch of an epoch), the
 * count travels in the same collective:
 *   1. each rank runs lsq_hip_backward_* with use_grad_scaling = 0, grad_scaler = 1 (unscaled terms) and dsdb_wide =
 *      packed, a double[2*channels + 1] whose last slot the caller sets to the shard's element count;
 *   2. ONE all-reduce(SUM) of `packed` (RCCL over xGMI; 24 bytes for a per-tensor quantizer);
 *   3. lsq_hip_sharded_finish derives the scaler from packed[2*channels] ON THE DEVICE (no host round trip) with the
 *      reference's precision chain for `dtype`'s arithmetic type, using p->quant_max, p->use_grad_scaling and
 *      p->grad_scaler, multiplies the fp64 sums once and rounds once: ds[c] = T(packed[c] * gs), db[c] = T(packed[C+c] * gs).
 * The result differs from the up-front route only in where the scaler's rounding happens (per term there, once per sum
 * here): well inside the 1e-6 parity budget.  A global count of 0 stores zeros.  per_channel = 0 requires channels = 1. */
int lsq_hip_sharded_finish(int dtype, const double* packed, int64_t channels, int32_t per_channel, const lsq_params* p,
                           void* ds, void* db, void* stream);

/* ---- many per-channel quantizers in one launch --------------------------------------------------- */

/* The weight quantizers of a QAT model: one lsq_forward_per_channel / lsq_backward_per_channel pair per conv / linear layer
 * and step in the reference (quantized/modules/observers.py:458-461 through lsq.cpp:104-134), each launch-latency-bound on a
 * tensor of a few MB.  These entry points take the tensors of many such calls at once and put their channels into ONE grid
 * (one launch per 32 tensors; no workspace, no finalize launch, HIP-graph capturable).
 * `items` is a HOST array (read before the call returns); every pointer in it is a device pointer with the meaning of the
 * same-named argument of the single-tensor entry points: the forward reads x, scale, shift and writes y; the backward reads
 * grad, x, scale, shift and writes dx, ds[channels], db[channels].  All tensors share `dtype` and `p` (numel_for_scaler must
 * be <= 0: every tensor's gradient scaler uses its own element count).  Every tensor must satisfy
 * lsq_hip_per_channel_multi_ok -- the tensors the single-tensor launch policy walks with one workgroup per channel
 * (packet-aligned channel rows of at least an eighth of a workgroup's 4 KiB span, few outer indices: conv / linear weights
 * quantized along axis 0); results are then bit-identical
 * to the single-tensor calls (the same walk and summation order).  Other tensors go through the single-tensor entry points. */
typedef struct lsq_pc_item {
    const void* x;
    const void* grad; /* backward only */
    void* y;          /* forward only */
    void* dx;         /* backward only */
    const void* scale;
    const void* shift;
    void* ds; /* backward only */
    void* db; /* backward only */
    int64_t outer, channels, inner;
} lsq_pc_item;

/* 1 if a dense [outer, channels, inner] tensor of `dtype` may take part in a multi-tensor launch (aligned16: all its
 * buffers are 16-byte aligned), 0 otherwise.  Host-only; depends on the current device's CU count. */
int lsq_hip_per_channel_multi_ok(int dtype, int64_t outer, int64_t channels, int64_t inner, int aligned16);

int lsq_hip_forward_per_channel_multi(int dtype, const lsq_pc_item* items, int32_t count, const lsq_params* p, void* stream);
int lsq_hip_backward_per_channel_multi(int dtype, const lsq_pc_item* items, int32_t count, const lsq_params* p, void* stream);

/* ---- which launch a per-channel backward gets (host-only, nothing is launched) ---------------------- */

/* The launch policy of lsq_hip_backward_per_channel run as a PLAN for a dense [outer, channels, inner] tensor of `dtype`
 * whose buffers are (aligned16 != 0) or are not 16-byte aligned, with the modes of `p` (sym / init_mode / eval_mode):
 *   out8 = [grid x, grid y, workgroups resident per CU the grid was sized for (0: not used), registers of the kernel
 *           instantiation (0: not asked), kernel family -- 1 = 256-lane windows, 2 = row-group windows, 3 = segment walk,
 *           4 = owner windows (one launch, no workspace) --, LDS-DMA ring depth (0 = register loops), workgroup size,
 *           ring copies issued with the streaming hint]
 * It is the same code path the launch takes (the sizes of lsq_hip_backward_per_channel_workspace come from it too), so a
 * test on the SHIPPED library can tell which kernel family a shape runs without a debug build (tests/test_shipped_binary_gpu.py).
 * Depends on the current device (CU count, the instantiation's register count): needs a GPU. */
int lsq_hip_plan_backward_per_channel(int dtype, int64_t outer, int64_t channels, int64_t inner, int aligned16,
                                      const lsq_params* p, int32_t* out8);

/* ---- rank communicator: the one collective of the batch-sharded backward ------------------------------- */

/* north_star: "batches shard across the 8 GPUs of one node with a single RCCL all-reduce over xGMI for the scale/shift
 * gradient scalars".  The reference has no distributed code; a data-parallel user of it would all-reduce scale.grad / shift.grad
 * through torch.distributed, whose enqueue costs ~60 us of host time per call -- as much as a rank's whole config-4 step
 * takes on the GPU.  These entry points issue the same RCCL all-reduce directly, from the library: a communicator of its own
 * (one per process group and rank, created once from a 128-byte id that rank 0 generates and the host layer broadcasts),
 * and two ways to run a reduction --
 *   lsq_hip_comm_all_reduce        on `stream`, in order with the kernels around it;
 *   lsq_hip_comm_all_reduce_begin  on the communicator's own stream, after everything enqueued on `stream`
 *                                  so far; `stream` itself goes on (the next step's kernels overlap the reduction) until
 *   lsq_hip_comm_all_reduce_end    makes `stream` wait for reduction `ticket`, or lsq_hip_comm_join for all of them.
 * Both forms are legal under HIP-graph capture when every begin is ended before the capture ends.  send == recv reduces in
 * place.  Elements: LSQ_F64 (the un-rounded [sum ds, sum db, count] of lsq_hip_backward_*'s dsdb_wide) or LSQ_F32 (the packed
 * [min, -max] of the rank-synchronised observer).  The calls are made on the device the communicator was created on (the
 * current device of lsq_hip_comm_create), from one thread at a time per communicator, in the same order on every rank.
 * RCCL is resolved with dlopen at the first call (the copy the process already has loaded, else the system's): liblsq_hip.so
 * has no link-time dependency on it, and without RCCL these calls return LSQ_ECOMM. */
typedef struct lsq_comm lsq_comm;
#define LSQ_COMM_ID_BYTES 128
enum lsq_comm_op { LSQ_COMM_SUM = 0, LSQ_COMM_MIN = 1, LSQ_COMM_MAX = 2 };

/* What a caller may choose about a communicator (NULL wherever it is taken = the defaults, all zero).
 *   size                 sizeof(lsq_comm_options) of the caller's build (the struct may grow at the end);
 *   event_system_fence   0: the events that order a begun reduction against the caller's stream are recorded WITHOUT the
 *                        system-scope fence (hipEventDisableSystemFence) -- producer, reduction and consumer run on one device,
 *                        where the agent-scope release every kernel ends with is enough, and the default fence (write back +
 *                        invalidate the L2s) costs a BASELINE-config-4 shard step 2-8 us (profiles/r05_comm_cost.txt);
 *                        1: default events.  The host layer is expected to CHECK the unfenced form on the transport at hand
 *                        before relying on it (torchlsq.distributed.native_comm: 128 reductions of changing values through
 *                        begin / side stream / join) and to switch with lsq_hip_comm_configure when the check fails. */
typedef struct lsq_comm_options {
    int32_t size;
    int32_t event_system_fence;
    int32_t reserved[2];
} lsq_comm_options;

int lsq_hip_comm_unique_id(void* id /* LSQ_COMM_ID_BYTES, host memory */);
/* SETUP, collective: returns once every rank of `nranks` has joined with the same id (RCCL's bootstrap: blocks, allocates). */
int lsq_hip_comm_create(const void* id, int32_t rank, int32_t nranks, const lsq_comm_options* options, lsq_comm** out);
/* SETUP: waits for the communicator's own stream to drain, then applies `options` (today: swaps the event set).  Not between a
 * begin and its end / join. */
int lsq_hip_comm_configure(lsq_comm* comm, const lsq_comm_options* options);
/* SETUP: picks, by measurement against `stream` (the stream the caller computes on), which of the communicator's candidate
 * streams the overlapped reductions run on, so that it does not share a hardware queue with `stream` (a cross-stream wait
 * parked in the compute stream's queue stalls its next kernel 13-22 us: profiles/r05_comm_cost.txt).  Allocates up to
 * 256 MB of scratch (freed before it returns), synchronises `stream` and the candidates, ~1 ms.  Call it once, after
 * lsq_hip_comm_create, at the same point on every rank, before the first begin; LSQ_EINVAL while `stream` is capturing.
 * Without it the first candidate is used -- correct, possibly on the compute stream's queue.  A box too short of memory for
 * the measurement keeps the first candidate and returns LSQ_OK (lsq_hip_comm_info says which). */
int lsq_hip_comm_tune(lsq_comm* comm, void* stream);
/* SETUP, collective (RCCL's teardown).  The communicator's stream is parked for the next communicator of the device, not
 * destroyed (a host allocator may still hold it as the consumer of a buffer). */
int lsq_hip_comm_destroy(lsq_comm* comm);
/* out8 = [rank, nranks, device, side-stream choice, event_system_fence, RCCL version code, reductions begun so far (mod 2^31),
 *         ticket ring size]; side-stream choice: 0 = lsq_hip_comm_tune has not run, 2 = it ran and no candidate was measurably
 * apart from the caller's stream (or the measurement could not be made): the first is used, 2 + k = candidate k - 1 was chosen */
int lsq_hip_comm_info(const lsq_comm* comm, int32_t* out8);
/* The communicator's own stream (a hipStream_t), for work that CONSUMES a reduction begun with lsq_hip_comm_all_reduce_begin
 * without making the caller's stream wait for it: enqueue the consumer (lsq_hip_sharded_finish, a cast) on this stream behind
 * the begin, and join the caller's stream once per training step -- lsq_hip_comm_join -- instead
 * of once per reduction (a cross-stream wait costs the GPU ~7 us each way, profiles/r05_comm_cost.txt).  It changes only in
 * lsq_hip_comm_tune. */
void* lsq_hip_comm_side_stream(const lsq_comm* comm);
/* `stream` waits for EVERYTHING enqueued on the communicator's stream so far -- the reductions begun and whatever consumers
 * the caller put behind them there: the once-per-step join of the pattern above. */
int lsq_hip_comm_join(lsq_comm* comm, void* stream);
int lsq_hip_comm_all_reduce(lsq_comm* comm, const void* send, void* recv, int64_t count, int dtype, int op, void* stream);
/* Tickets come from a ring of 8: begin number n + 8 re-records the events of begin n.  A late lsq_hip_comm_all_reduce_end on
 * the old ticket then waits for the NEWER reduction as well -- it over-waits, it never under-waits (the communicator's stream is
 * in order) -- so more than 8 outstanding begins are legal; lsq_hip_comm_join needs no ticket at all. */
int lsq_hip_comm_all_reduce_begin(lsq_comm* comm, const void* send, void* recv, int64_t count, int dtype, int op,
                                  void* stream, int32_t* ticket);
int lsq_hip_comm_all_reduce_end(lsq_comm* comm, int32_t ticket, void* stream);

/* ---- a gradient in another dense memory order than its input -------------------------------------------- */

/* dst[a][c][b] = src[a][b][c] for dense arrays of `dtype` elements (a < A, b < B, c < C): the batched transposition that
 * takes a contiguous NCHW gradient into the memory order of a channels-last input (A = N, B = C, C = H*W) or back (B = H*W,
 * C = C).  The reference's backward meets grad and x element by element through one TensorIterator whatever their strides
 * (lsq_cpu.cpp:80-90, :229-249); these kernels walk x's memory order, so the host layers bring such a grad into it first --
 * with this pass (LDS tiles, both sides coalesced) instead of the framework's generic strided copy (2.5 x slower on MI355X,
 * profiles/r06_layout_workloads.txt).  src and dst must not overlap; element-aligned pointers. */
int lsq_hip_relayout(int dtype, const void* src, void* dst, int64_t a, int64_t b, int64_t c, void* stream);

/* ---- eval-mode backward from the saved mask ---------------------------------------------------- */

/* dx = grad * mask for the eval-mode / plain fake-quantizer backward (lsq_kernel.h:126-145: dX only,
 * d_scale = d_shift = 0), per-tensor and per-channel alike: reads grad + 1 byte per element instead of
 * grad + x (9 instead of 12 bytes per fp32 element) and lets autograd keep the 1-byte mask instead of x.
 * `mask` is the aux_kind = 1 output of the forward, same memory order as grad / dx. */
int lsq_hip_backward_from_mask(int dtype, const void* grad, const void* mask, void* dx, int64_t n, void* stream);

/* ---- observer statistics (init phase) ------------------------------------------------------ */

/* Running min / max of a tensor in ONE read-only pass: what the torch MinMax observers that the
 * reference module runs before the fake-quantize op during its initialisation batches compute with
 * torch.aminmax (reference quantized/modules/observers.py:446-449).  Semantics of torch.aminmax: exact
 * min and max; a NaN anywhere (in the channel) makes both results NaN.
 * min_out / max_out: 1 element (per-tensor) or `channels` elements, float for F32|BF16|F16 storage,
 * double for F64.  The workspace size comes from lsq_hip_minmax_workspace (per-tensor: channels = 1,
 * outer = 1, inner = n). */
size_t lsq_hip_minmax_workspace(int dtype, int64_t outer, int64_t channels, int64_t inner);
int lsq_hip_minmax_per_tensor(int dtype, const void* x, int64_t n, void* min_out, void* max_out,
                              void* workspace, size_t workspace_bytes, void* stream);
int lsq_hip_minmax_per_channel(int dtype, const void* x, int64_t outer, int64_t channels, int64_t inner,
                               void* min_out, void* max_out, void* workspace, size_t workspace_bytes,
                               void* stream);

/* Mean and (unbiased) standard deviation in ONE read-only pass: the statistics behind the 3-sigma initialisation
 * of a weight quantizer's scale, scale = max(|mu - 3 sigma|, |mu + 3 sigma|) / 2^bits, which the reference module
 * computes with torch.mean / torch.std over the tensor or per channel over the other axes on the first call
 * (reference quantized/modules/observers.py:329-337).  fp64 shifted-data sums (pivot = the channel's first
 * element); n = 1 gives std = NaN and non-finite inputs propagate, as in torch.  Outputs and workspace as for
 * lsq_hip_minmax_* (float for F32|BF16|F16 storage, double for F64). */
size_t lsq_hip_meanstd_workspace(int dtype, int64_t outer, int64_t channels, int64_t inner);
int lsq_hip_meanstd_per_tensor(int dtype, const void* x, int64_t n, void* mean_out, void* std_out,
                               void* workspace, size_t workspace_bytes, void* stream);
int lsq_hip_meanstd_per_channel(int dtype, const void* x, int64_t outer, int64_t channels, int64_t inner,
                                void* mean_out, void* std_out, void* workspace, size_t workspace_bytes,
                                void* stream);

/* ---- observer-driven initialisation: the tail after the statistics pass ------------------------------ */

/* ONE launch for what the reference module does with a dozen tiny tensor operations and several device
 * synchronisations after every min/max pass of its initialisation batches (reference
 * quantized/modules/observers.py:446-449: observer update, calculate_qparams, _set_weights):
 *   1. fold the batch's per-channel min / max into the observer's running state, in place:
 *        mode 1  running min / max         (torch MinMaxObserver / PerChannelMinMaxObserver)
 *        mode 2  moving average            state += averaging_constant * (current - state)
 *      `first`: 1 = the state is uninitialised, take the batch values; 0 = update; -1 = decide per element on the
 *      device (state still at +inf / -inf: what the per-tensor torch observers test on the host);
 *   2. derive (scale, zero_point) from the state exactly as torch's UniformQuantizationObserverBase._calculate_qparams
 *      does (affine: scale = (max(max,0) - min(min,0)) * (1 / (quant_max - quant_min)), zero_point = clamp(quant_min -
 *      rne(min / scale)); symmetric: scale = max(-min, max) * (1 / ((quant_max - quant_min) / 2)), zero_point =
 *      `zero_point_symmetric`; scale >= eps), every operation individually rounded like the tensor operations it replaces;
 *   3. store the LSQ parameters: scale_out = scale, shift_out = float(-zero_point) * scale (reference :346-373).
 * fp32 state and parameters, `channels` of each (1 for per-tensor observers). */
typedef struct lsq_observer_update {
    int32_t mode;
    int32_t first;
    float averaging_constant;
    int32_t quant_min;
    int32_t quant_max;
    int32_t symmetric;
    int32_t zero_point_symmetric;
    float eps;
} lsq_observer_update;

int lsq_hip_observer_update(int64_t channels, const float* cur_min, const float* cur_max, float* min_state,
                            float* max_state, const lsq_observer_update* u, float* scale_out, float* shift_out,
                            void* stream);

#ifdef __cplusplus
}
#endif
#endif /* LSQ_HIP_H_ */
