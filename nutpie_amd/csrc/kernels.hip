// kernels.hip — the NUTS hot path as HIP kernels for gfx950 (CDNA4).
//
// One chain is owned by W wavefronts (W=1: one wavefront per chain, four independent chains per
// 256-thread workgroup; W>1: one workgroup of W waves per chain, cross-wave reductions through
// LDS).  A chain is a small state machine (engine_types.h: Phase) that runs leapfrogs, the
// iterative (non-recursive) NUTS tree, dual averaging and the diagonal mass-matrix update
// entirely on the device.  Lane l of a wave owns element pairs (128c + 2l, 128c + 2l + 1): every
// vector access is one 16-byte load/store per lane = 1 KiB contiguous per wave instruction.
//
// What this replaces (reference): nuts-rs' per-chain CPU worker that nutpie starts through
// `nuts_rs::Sampler::new` (src/wrapper.rs:977-1085) and feeds through `CpuLogpFunc::logp`
// (src/pymc.rs:197-215, src/stan.rs:454-463, src/pyfunc.rs:206-230).  Algorithm: SURVEY.md
// Appendix A; the numbers are defined by include/nphip_spec.h.  The tree here is NOT the crate's
// recursion: leaves are numbered inside a doubling and the (p, rho) summaries needed by later
// U-turn checks are written once, by the leapfrog itself, into a slot that is a pure function of
// the leaf number (engine_types.h), so no state is ever copied.
//
// MFMA is unused on purpose: the path is axpy/dot (0.2 flop/byte), bounded by memory bandwidth.
#include <hip/hip_runtime.h>

#include "../../include/nphip_spec.h"
#include "engine_types.h"

// address-space qualifiers only exist in the device pass (the host pass merely parses the kernels)
#if defined(__HIP_DEVICE_COMPILE__)
#define NPHIP_GLOBAL __attribute__((address_space(1)))
#define NPHIP_LDS __attribute__((address_space(3)))
#define NPHIP_CONST __attribute__((address_space(4)))
#else
#define NPHIP_GLOBAL
#define NPHIP_LDS
#define NPHIP_CONST
#endif

// Runtime-compiled device densities (nutpie_amd/density.py -> nphip_model_jit_density): this file is compiled once more, per
// model, with -DNPHIP_JIT_DENSITY -DNPHIP_PART=7 behind a generated prelude that defines `struct NphipData` and
//     __device__ double nphip_density(const NphipData& data, int dim, const double* x, double* grad, double* lds, const double* shared, int lane)
// — evaluated by ONE converged wavefront: x[dim] is the position row (global memory), grad[dim] receives the gradient, `lds`
// is the wave's private LDS scratch, `shared` LDS common to the chains of the workgroup, the return value the log-density
// (the same in every lane; non-finite = a recoverable error, src/pyfunc.rs:218-220) — and
//     __device__ void nphip_density_stage(const NphipData& data, double* shared, int thread, int n_threads)
// called once per workgroup and launch by all of its threads to fill `shared` (typically: the model's data, so that the
// density reads LDS instead of waiting out an L2 latency per access on a lone wave).  The kernel built from it is the REMOTE
// machine — the register-resident leaf with the evaluation as a call in its middle — with the host rendezvous replaced by
// that call.
#ifdef NPHIP_JIT_DENSITY
#define NPHIP_JIT 1
#else
#define NPHIP_JIT 0
#endif

namespace nphip {

// ----------------------------------------------------------------------------------------
// wave / workgroup primitives
// ----------------------------------------------------------------------------------------

template <int W>
// The register kernels run one wave per SIMD with (almost) the whole register file live.  Left alone, the
// scheduler hoists every load of the next phase above the current one and drives the allocator into scratch;
// a fence between the phases of a leaf keeps each phase's temporaries local to it.
#define NPHIP_PHASE_FENCE() __builtin_amdgcn_sched_barrier(0)
// lean kernels: the unrolled chunk sweeps must stay sweeps — without a fence per chunk the scheduler issues the loads of
// all chunks first and the allocator spills the resident state to make room for them
#define NPHIP_CHUNK_FENCE(k) __builtin_amdgcn_sched_barrier(0)
// one wave per chain: up to this many chunks per lane the kernel is built for two waves per SIMD (256 registers each).
// Not the kernel of a runtime-compiled density: its workgroup's LDS (four chains' scratch + the model's shared block) leaves one
// workgroup per CU anyway, and the density wants the registers — radon, 512 / 2048 chains, same box: 56.6 -> 62.1 / 96.2 -> 107.7 M
// leapfrogs/s for the traced torch density, 50.3 -> 51.5 / 90.6 -> 93.7 for the one written as expressions (profiles/r5_jit_occupancy.txt);
// a small density stays below 256 registers by itself.
#ifndef NPHIP_W1_OCC2_MAX
#if NPHIP_JIT
#define NPHIP_W1_OCC2_MAX 0
#else
#define NPHIP_W1_OCC2_MAX 3
#endif
#endif
// register kernels with several waves per chain: up to this many chunks per wave run two waves per SIMD (256 registers each)
#ifndef NPHIP_RW_OCC2_MAX
#define NPHIP_RW_OCC2_MAX 4
#endif
#ifndef NPHIP_CB_OCC
// waves per SIMD the launch-per-evaluation (callback) kernels are compiled for.  Two: 256 registers per lane, (almost) nothing
// spilled.  Measured at four waves per SIMD (128 registers, every chain of a 1024-chain batch x 4 waves on the device at once):
// 82 spilled VGPRs, eleven of them stored by every wave of every launch — 23 MB of scratch traffic per launch in a kernel that
// is bound by its memory traffic — 38.4 us per launch against 33.5 (profiles/r4_callback_kernels.txt)
#define NPHIP_CB_OCC(W) 2
#endif
#ifndef NPHIP_LEAN_OCC
// waves per SIMD of the lean kernels: 8 waves = one chain per CU at 256 VGPRs per wave (4 waves: 512 = VGPRs + AGPRs)
#define NPHIP_LEAN_OCC(W) ((W) <= 4 ? 1 : ((W) <= 8 ? 2 : 4))
#endif

__device__ __forceinline__ void chain_sync() {
    // make this chain's global stores visible to all of its lanes/waves
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    if (W > 1) __syncthreads();
}

// ---- wave reduction in the contract order of nphip_spec.h (DPP, no LDS traffic) ---------------------
// stage partners: l^1, l^2, mirror within 8, mirror within 16, l^16, then lanes {0, 32}.
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double x) {
    int lo = __double2loint(x), hi = __double2hiint(x);
    // every lane is active and every source lane of these permutations exists: no "old" value to tie the
    // destination to (saves the two register copies per stage update_dpp(old = x, ...) costs)
    lo = __builtin_amdgcn_mov_dpp(lo, CTRL, 0xF, 0xF, true);
    hi = __builtin_amdgcn_mov_dpp(hi, CTRL, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double swz16_f64(double x) {  // value of lane l ^ 16
    int lo = __builtin_amdgcn_ds_swizzle(__double2loint(x), 0x401F);
    int hi = __builtin_amdgcn_ds_swizzle(__double2hiint(x), 0x401F);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double readlane_f64(double x, int l) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(x), l), __builtin_amdgcn_readlane(__double2loint(x), l));
}
// value of lane l-1 (lane 0 receives `edge`) / lane l+1 (lane 63 receives `edge`): DPP wave shifts
__device__ __forceinline__ double wave_shr1(double x, double edge) {
    int lo = __builtin_amdgcn_update_dpp(__double2loint(edge), __double2loint(x), 0x138, 0xF, 0xF, false);
    int hi = __builtin_amdgcn_update_dpp(__double2hiint(edge), __double2hiint(x), 0x138, 0xF, 0xF, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_shl1(double x, double edge) {
    int lo = __builtin_amdgcn_update_dpp(__double2loint(edge), __double2loint(x), 0x130, 0xF, 0xF, false);
    int hi = __builtin_amdgcn_update_dpp(__double2hiint(edge), __double2hiint(x), 0x130, 0xF, 0xF, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_sum(double v) {
    v = v + dpp_f64<0xB1>(v);    // quad_perm [1,0,3,2]  : l ^ 1
    v = v + dpp_f64<0x4E>(v);    // quad_perm [2,3,0,1]  : l ^ 2
    v = v + dpp_f64<0x141>(v);   // row_half_mirror      : 7 - l  (within 8)
    v = v + dpp_f64<0x140>(v);   // row_mirror           : 15 - l (within 16)
    v = v + swz16_f64(v);        // l ^ 16
    return readlane_f64(v, 0) + readlane_f64(v, 32);
}

// Sum N values over the chain: wave_sum per value, then (W > 1) wave totals in wave order through LDS.
// N independent wave sums, stage by stage: the same operations per value as wave_sum (same bits), but consecutive instructions
// belong to different values — a lone wave otherwise waits out the latency of every DPP move -> add -> DPP move chain
// (measured on the 1000-dim kernel: 860 cycles per 4-value reduction issued value by value).
template <int N>
__device__ __forceinline__ void wave_sumN(double (&v)[N]) {
#pragma unroll
    for (int n = 0; n < N; ++n) v[n] = v[n] + dpp_f64<0xB1>(v[n]);    // l ^ 1
#pragma unroll
    for (int n = 0; n < N; ++n) v[n] = v[n] + dpp_f64<0x4E>(v[n]);    // l ^ 2
#pragma unroll
    for (int n = 0; n < N; ++n) v[n] = v[n] + dpp_f64<0x141>(v[n]);   // 7 - l  (within 8)
#pragma unroll
    for (int n = 0; n < N; ++n) v[n] = v[n] + dpp_f64<0x140>(v[n]);   // 15 - l (within 16)
#pragma unroll
    for (int n = 0; n < N; ++n) v[n] = v[n] + swz16_f64(v[n]);        // l ^ 16
#pragma unroll
    for (int n = 0; n < N; ++n) v[n] = readlane_f64(v[n], 0) + readlane_f64(v[n], 32);
}

// TWO, FOUR or EIGHT wave sums with the lanes halved along the way (round 5): the same additions on the same operands as
// wave_sumN — the bits of the contract's order — in 29 / 50 / 95 instructions instead of 40 / 80 / 160 (described for four).  After stage 1 a register holds value 0's pair sums in the even lanes and
// value 1's in the odd ones (X; Y likewise for values 2, 3); after stage 2 lane 4i + c holds the sum of value c over quad i (Z).  From there
// only the lanes that end in the result are kept up: the contract adds, per row of 16 lanes, (q0 + q1) + (q3 + q2) — in the butterfly every lane
// of a quad holds the quad's sum, so lane 0's mirror partners 7, 15 stand for quads 1, 3 —, then row 0 + row 1, then the two halves of the wave.
// (The selections by lane parity are v_cndmask: DPP's bank mask selects quads of lanes, not lanes of a quad.)
// Here B = Z + shr4(Z) has q1 + q0 in lanes 4..7 and q3 + q2 in lanes 12..15 (IEEE addition commutes bit for bit), C = B + shr8(B) has
// (q3 + q2) + (q1 + q0) in lanes 12..15, D = C + swizzle16(C) row 0 + row 1 there, and the result is lane 12 + c of the lower half plus lane
// 44 + c of the upper one.
__device__ __forceinline__ double swz4_f64(double x) {  // value of lane l ^ 4
    int lo = __builtin_amdgcn_ds_swizzle(__double2loint(x), 0x101F);
    int hi = __builtin_amdgcn_ds_swizzle(__double2hiint(x), 0x101F);
    return __hiloint2double(hi, lo);
}
// lower half of the wave + upper half, in every lane (the contract's last addition: lane 0 + lane 32): v_permlane32_swap (gfx950) exchanges lanes
// 32..63 of its first operand with lanes 0..31 of its second — given the same value twice it returns the lower half in both halves and the
// upper half in both halves
__device__ __forceinline__ double halves_sum_f64(double x) {
#ifndef NPHIP_NO_PERMLANE_SWAP
    const int lo = __double2loint(x), hi = __double2hiint(x);
    const auto r0 = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
    const auto r1 = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    return __hiloint2double(r1[0], r0[0]) + __hiloint2double(r1[1], r0[1]);
#else
    return x;
#endif
}
// one halving step: lanes whose bit `sel` is clear go on with a, the others with b; the partner (CTRL) hands over what this lane does not keep
template <int CTRL>
__device__ __forceinline__ double halve(bool sel, double a, double b) {
    const double keep = sel ? b : a, give = sel ? a : b;
    return keep + dpp_f64<CTRL>(give);
}
template <int N>
__device__ __forceinline__ void wave_sumN_halving(double (&v)[N]) {
    static_assert(N == 2 || N == 4 || N == 8, "");
    const unsigned lane_ = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    const bool b0 = (lane_ & 1u) != 0u, b1 = (lane_ & 2u) != 0u, b2 = (lane_ & 4u) != 0u;
    if constexpr (N == 2) {
        double X = halve<0xB1>(b0, v[0], v[1]);          // even lanes: value 0, odd lanes: value 1
        X = X + dpp_f64<0x4E>(X);                          // l ^ 2 keeps the parity
        const double B = X + dpp_f64<0x114>(X), C_ = B + dpp_f64<0x118>(B), D_ = C_ + swz16_f64(C_);
#ifndef NPHIP_NO_PERMLANE_SWAP
        const double T_ = halves_sum_f64(D_);
        v[0] = readlane_f64(T_, 12);
        v[1] = readlane_f64(T_, 13);
#else
        v[0] = readlane_f64(D_, 12) + readlane_f64(D_, 44);
        v[1] = readlane_f64(D_, 13) + readlane_f64(D_, 45);
#endif
    } else if constexpr (N == 4) {
        const double X = halve<0xB1>(b0, v[0], v[1]), Y = halve<0xB1>(b0, v[2], v[3]);
        const double Z = halve<0x4E>(b1, X, Y);            // lane 4i + c: value c summed over quad i
        const double B = Z + dpp_f64<0x114>(Z), C_ = B + dpp_f64<0x118>(B), D_ = C_ + swz16_f64(C_);
#ifndef NPHIP_NO_PERMLANE_SWAP
        const double T_ = halves_sum_f64(D_);
#pragma unroll
        for (int c = 0; c < 4; ++c) v[c] = readlane_f64(T_, 12 + c);
#else
#pragma unroll
        for (int c = 0; c < 4; ++c) v[c] = readlane_f64(D_, 12 + c) + readlane_f64(D_, 44 + c);
#endif
    } else {
        const double X0 = halve<0xB1>(b0, v[0], v[1]), Y0 = halve<0xB1>(b0, v[2], v[3]);
        const double X1 = halve<0xB1>(b0, v[4], v[5]), Y1 = halve<0xB1>(b0, v[6], v[7]);
        const double Za = halve<0x4E>(b1, X0, Y0), Zb = halve<0x4E>(b1, X1, Y1);
        // stage 3: the contract's partner 7 - l holds the OTHER quad's sum of the same value, as lane l ^ 4 does here
        const double keep = b2 ? Zb : Za, give = b2 ? Za : Zb;
        const double Wv = keep + swz4_f64(give);           // lanes 8i + c: values 0..3 over eight lanes, 8i + 4 + c: values 4..7
        const double C_ = Wv + dpp_f64<0x118>(Wv), D_ = C_ + swz16_f64(C_);
#ifndef NPHIP_NO_PERMLANE_SWAP
        const double T_ = halves_sum_f64(D_);
#pragma unroll
        for (int c = 0; c < 8; ++c) v[c] = readlane_f64(T_, 8 + c);
#else
#pragma unroll
        for (int c = 0; c < 8; ++c) v[c] = readlane_f64(D_, 8 + c) + readlane_f64(D_, 40 + c);
#endif
    }
}

// HALVE = false: the plain butterfly (the lean kernels: their register file is full, the selects of the halving steps cost them 48 more bytes of
// scratch per lane and 4 % more memory traffic — measured: 32.6 -> 36.1 ms per launch at D = 10 000)
template <int W, int N, bool TRAILING_BARRIER = true, bool HALVE = true>
__device__ __forceinline__ void reduceN(double (&v)[N], NPHIP_LDS double* red) {
#ifndef NPHIP_NO_HALVING_SUM
    if constexpr (HALVE && (N == 2 || N == 4 || N == 8)) wave_sumN_halving(v); else wave_sumN(v);
#else
    wave_sumN(v);
#endif
    if (W > 1) {
        const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
        if (lane == 0) {
#pragma unroll
            for (int n = 0; n < N; ++n) red[N * wave + n] = v[n];
        }
        __syncthreads();
#pragma unroll
        for (int n = 0; n < N; ++n) {
            double t = red[n];
            for (int w = 1; w < W; ++w) t = t + red[N * w + n];
            v[n] = t;
        }
        if (TRAILING_BARRIER) __syncthreads();   // (callers that alternate between two scratch areas do not need it)
    }
}
template <int W>
__device__ __forceinline__ void reduce2(double& a, double& b, NPHIP_LDS double* red) {
    double v[2] = {a, b};
    reduceN<W, 2>(v, red);
    a = v[0]; b = v[1];
}

// Explicit address spaces: the engine's pointers arrive inside a by-value struct, so clang cannot infer
// that they are global and would emit flat_load/flat_store (slower, and they tie vmcnt to lgkmcnt).
typedef NPHIP_LDS Ctl* LdsCtl;
typedef NPHIP_LDS double* LdsDouble;

__device__ __forceinline__ double2 ld2(const double* p, int64_t i) { return *(const NPHIP_GLOBAL double2*)(p + i); }
__device__ __forceinline__ void st2(double* p, int64_t i, double2 v) { *(NPHIP_GLOBAL double2*)(p + i) = v; }
// a pair of single-precision values as doubles (the columns of the low-rank metric: fp32 in memory, every operation on them fp64)
__device__ __forceinline__ double2 ld2f(const float* p, int64_t i) {
    const float2 f = *(const NPHIP_GLOBAL float2*)(p + i);
    double2 v;
    v.x = (double)f.x; v.y = (double)f.y;
    return v;
}
__device__ __forceinline__ double ld1(const double* p, int64_t i) { return *(const NPHIP_GLOBAL double*)(p + i); }
__device__ __forceinline__ void st1(double* p, int64_t i, double v) { *(NPHIP_GLOBAL double*)(p + i) = v; }
// the same with a 32-bit byte offset per lane: with a uniform base the access is `global_load ... v_off, s[base:base+1]` — ONE
// address register per lane for every vector of a pass instead of a 64-bit pair per (vector, chunk)
__device__ __forceinline__ double2 ld2b(const double* p, uint32_t off) { return *(const NPHIP_GLOBAL double2*)((const NPHIP_GLOBAL char*)p + off); }
__device__ __forceinline__ void st2b(double* p, uint32_t off, double2 v) { *(NPHIP_GLOBAL double2*)((NPHIP_GLOBAL char*)p + off) = v; }
// dense rows (ld == dim, possibly odd / unaligned): guarded scalar accesses
// (an even row length — rows start on 16-byte boundaries then, and no pair straddles the end — takes whole pairs)
__device__ __forceinline__ double2 ld2_dense(const double* p, int64_t i, int64_t D) {
    double2 v;
    if ((D & 1) == 0) {
        v.x = 0.0; v.y = 0.0;
        if (i < D) v = ld2(p, i);
        return v;
    }
    v.x = (i < D) ? ld1(p, i) : 0.0;
    v.y = (i + 1 < D) ? ld1(p, i + 1) : 0.0;
    return v;
}
__device__ __forceinline__ void st2_dense(double* p, int64_t i, int64_t D, double2 v) {
    if ((D & 1) == 0) {
        if (i < D) st2(p, i, v);
        return;
    }
    if (i < D) st1(p, i, v.x);
    if (i + 1 < D) st1(p, i + 1, v.y);
}
// all of this wave's memory operations are complete (gfx9 encoding: vmcnt(0), expcnt / lgkmcnt untouched)
__device__ __forceinline__ void wait_vm0() {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_waitcnt(0x0F70);
    asm volatile("" ::: "memory");
}
__device__ __forceinline__ double clamp_mm(double v) { return v < 1e-20 ? 1e-20 : (v > 1e20 ? 1e20 : v); }

}  // namespace nphip
#include "dense_tile.h"   // the fp64 MFMA tile of the dense-precision Gaussian's gradient
namespace nphip {

// ---- dense-precision Gaussian, resident form: one evaluation ROUND of the whole launch ------------------------------------------
// The register-resident leaf keeps a chain's state in VGPRs and calls its evaluation in the middle (REMOTE).  For the dense Gaussian
// the evaluation of ALL chains is one GEMM, so the call is a rendezvous of workgroups: every wave has written its position row ->
// barrier -> each workgroup computes a 64 x 64 tile of G = -(X - mu) P on the matrix cores (dense_tile.h; four waves, 32 x 32 each)
// -> barrier -> every wave reads its gradient row back and goes on with its leaf.  No launch, no trip of the chain state through
// memory, no host.
// Who waits for whom: the GEMM of a row block needs only the positions of ITS chains, and a chain only its own row of G.  So the
// 16 workgroups whose 64 chains form one row block are a CLUSTER that synchronises among itself (two counters per cluster), and the
// clusters drift apart freely.  Workgroup b runs on XCD b % 8 (round-robin dispatch): cluster = (b % 8) + 8 * (b / 128), so a cluster's
// positions and gradients stay in ONE XCD's L2; member j = (b / 8) % 16 of m computes the column tiles j, j + m, ...
// Every wave of the launch takes part in every round (a chain that has finished, or a wave without a chain, as a ghost): the
// counters' targets are multiples of the cluster size.  A wait that lasts a second gives up, flags the launch and everybody leaves.
// WHO IS IN A CLUSTER is decided at run time (dg_register / dg_enroll, k_advance): every workgroup reads the XCC_ID of the die it was
// dispatched to and takes the next seat of that die; sixteen consecutive seats are a cluster.  All members of a cluster then share ONE L2
// — the coherence point for everything they exchange: a position or gradient row is visible to the other members once its store has
// been acknowledged (vmcnt), and a reader only has to drop its own CU's L1 (buffer_inv sc0: the workgroup-scope acquire of threadgroup-
// split mode).  The agent-scope alternative — an L2 write-back and an L2 invalidate around every rendezvous, which a cluster spanning dies
// needs — cost 20 us of a 98 us round, most of it by evicting the precision matrix from L2 every round (profiles/r6_dense_resident.txt).
// Per wave in LDS (g_dgwho + 24 * wave): [0] cluster (global index: die * 16 + cluster of the die), [1] member, [2] members, [3] die,
// [4] seat, [8 .. 8 + members) the members' workgroup ids (row r of the cluster's block is chain 4 * id[r / 4] + r % 4).
__shared__ int g_dgwho[4 * 24];
// (Measured and rejected, round 6 — profiles/r6_dense_resident.txt: touching the P-slot operands of the leaf's merge cascade ahead of time
//  (global_load_lds into a dump, one dword per 128-byte line).  A round waits for the slowest of a cluster's 64 leaves, and 4 % of the leaves take
//  more than 20 us — the deep cascades.  Issued after the round, in front of the leaf's own loads: 74.9 against 74.0 us per round (loads return in
//  order); issued when the wave has finished its GEMM tile: 78.8 (the wave's wait for its gradient stores now waits for them too).)
constexpr int kDgMaxClusters = 128;                    // 8 dies x 16 clusters of 16 workgroups
constexpr int kDgAbort = kDgMaxClusters * 32;          // word offsets into Args::dg_sync: the abort word (a line of its own)
constexpr int kDgSeats = kDgAbort + 16;                // seats taken per die [8] (32-bit)
constexpr int kDgTable = kDgSeats + 16;                // workgroup id of every seat [8][256] (32-bit)
constexpr int kDgWords = kDgTable + 8 * 256 / 2;       // (size of dg_sync in 64-bit words)
static_assert(kDgWords == kDgSyncWords, "engine_types.h: kDgSyncWords");
__device__ __forceinline__ int64_t dg_chain_of(const NPHIP_LDS int* who, int r) { return 4 * (int64_t)who[8 + (r >> 2)] + (r & 3); }
// kernel start, every thread of the workgroup: take a seat on this die (static: the seats a round-robin dispatch would give)
__device__ __forceinline__ void dg_register(const NPHIP_CONST Args& A) {
    if (threadIdx.x == 0) {
        unsigned die, seat;
        if (A.dg_variant & 64) { die = blockIdx.x & 7u; seat = blockIdx.x >> 3; }
        else {
            die = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 7u;   // hwreg(HW_REG_XCC_ID, 0, 4)
            seat = atomicAdd((unsigned*)(A.dg_sync + kDgSeats) + die, 1u);
        }
        __hip_atomic_store((unsigned*)(A.dg_sync + kDgTable) + die * 256u + seat, (unsigned)blockIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        for (int w = 0; w < 4; ++w) { g_dgwho[24 * w + 3] = (int)die; g_dgwho[24 * w + 4] = (int)seat; }
    }
    __syncthreads();
}
// after the roll call (every workgroup of the launch has taken its seat), every wave: who are the members of my cluster
__device__ __forceinline__ void dg_enroll(const NPHIP_CONST Args& A) {
    const int wib = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    NPHIP_LDS int* who = (NPHIP_LDS int*)g_dgwho + 24 * wib;
    const int die = who[3], seat = who[4];
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    int taken;
    if (A.dg_variant & 64) { const int G = (int)gridDim.x; taken = (G - die + 7) / 8; }
    else taken = (int)__hip_atomic_load((unsigned*)(A.dg_sync + kDgSeats) + die, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int cl = seat >> 4, left = taken - 16 * cl, m = left > 16 ? 16 : left;
    if (lane < 16) {
        int id = 0;
        if (lane < m) {
            if (A.dg_variant & 64) id = die + 8 * (16 * cl + lane);
            else id = (int)__hip_atomic_load((unsigned*)(A.dg_sync + kDgTable) + die * 256 + 16 * cl + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        who[8 + lane] = id;
    }
    if (lane == 0) { who[0] = die * 16 + cl; who[1] = seat & 15; who[2] = m; }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
}
__device__ __forceinline__ void dg_wait(unsigned long long* ctr, unsigned long long want, unsigned long long* abort_word) {
    const long long t0 = wall_clock64();   // 100 MHz
    while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
        __builtin_amdgcn_s_sleep(1);
        if (__hip_atomic_load(abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0ull) break;
        if (wall_clock64() - t0 > 100000000ll) { __hip_atomic_store(abort_word, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
    }
}
// prof (measurement, dg_variant & 32): this wave's cycle counters — [8] wait for the positions, [9] the GEMM, [10] wait for the gradients (shader
// cycles), [11] the whole round in 100 MHz ticks, [12] rounds
__device__ __forceinline__ void dg_round(const NPHIP_CONST Args& A, int round, int64_t n_chains, NPHIP_LDS int64_t* prof = nullptr) {
    const int wib = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    const NPHIP_LDS int* who = (const NPHIP_LDS int*)g_dgwho + 24 * wib;
    const int who_member = who[1], who_m = who[2];
    unsigned long long* arrive = A.dg_sync + (size_t)who[0] * 32;
    unsigned long long* done = arrive + 16;
    unsigned long long* abort_word = A.dg_sync + kDgAbort;
    const unsigned long long want = (unsigned long long)who_m * (unsigned long long)(round + 1);
    // How the exchanged rows become visible.  Writers: a plain store is in the die's L2 — the cache every member of the cluster shares — once it
    // is acknowledged (wait_vm0): no write-back.  Readers must not be served a stale line by their OWN CU's L1.  Measured on the device
    // (profiles/r6_dense_resident.txt): `buffer_inv sc0` does not drop it outside threadgroup-split mode (wrong gradients); `buffer_inv sc1` does,
    // but it also empties the die's L2 of everything it may not keep across an agent-scope acquire — the precision matrix included (+6 us of
    // GEMM per round).  So (default) ONE wave of the workgroup issues it, once per round, before the position rows are read; the gradient
    // rows are never read through the L1 at all: their one reader takes them by agent-scope loads (Machine::ge_ld).  dg_variant & 128
    // (A/B): every wave invalidates after both waits and the gradient rows are read by plain loads.
    const bool inv_both = (A.dg_variant & 128) != 0;
    wait_vm0();          // this wave's position row is on its way to L2
    __syncthreads();     // ... and so are the rows of the workgroup's other chains
    const int variant = A.dg_variant;
    const bool profiling = (variant & 32) && prof != nullptr;
    long long pc0 = 0, pw0 = 0, pc1 = 0, pc2 = 0;
    if (profiling) {
        pc0 = (long long)__builtin_readcyclecounter(); pw0 = wall_clock64();
        if (lane == 0 && prof[6] != 0) {   // the leaf between two rounds: < 5, < 10, < 15, < 20, < 30, >= 30 us
            const long long dt = pw0 - prof[6];
            prof[dt < 500 ? 0 : (dt < 1000 ? 1 : (dt < 1500 ? 2 : (dt < 2000 ? 3 : (dt < 3000 ? 4 : 5))))] += 1;
        }
    }
    if (wib == 0) {
        if (lane == 0) {
            __hip_atomic_fetch_add(arrive, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            dg_wait(arrive, want, abort_word);
        }
        if (!inv_both) asm volatile("buffer_inv sc1" ::: "memory");
    }
    __syncthreads();
    if (inv_both) asm volatile("buffer_inv sc1" ::: "memory");
    if (profiling) pc1 = (long long)__builtin_readcyclecounter();
    // ---- this workgroup's tiles: rows = the cluster's chains, columns = tiles member, member + m, ...
    const int64_t D = A.dim, KP = A.dg_KP;
    const int rh = wib >> 1, ch = wib & 1;
    const bool rows_live = 8 * rh < who_m;   // (a small cluster has no chains in the upper half of its block)
    if (rows_live) {
        const int64_t Nt = (D + 63) / 64;
        const double* xrow[2];
        int64_t crow[2];
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            int64_t cn = dg_chain_of(who, 32 * rh + 16 * r + (lane & 15));
            cn = (((32 * rh + 16 * r + (lane & 15)) >> 2) < who_m && cn < n_chains) ? cn : dg_chain_of(who, 0);    // (rows without a chain: a row that exists, never stored)
            crow[r] = cn;
            xrow[r] = A.qeval + (size_t)cn * D;
        }
        (void)crow;
        for (int64_t nt = who_member; nt < Nt; nt += who_m) {
            const int64_t c0 = nt * 64 + 32 * ch;
            if (c0 >= D) continue;
            const double* prow[2];
#pragma unroll
            for (int c = 0; c < 2; ++c) prow[c] = A.dg_P + (size_t)(c0 + 16 * c + (lane & 15)) * KP;
            dg_v4 acc[2][2];
            dense_block_32x32(xrow, prow, A.dg_mu, D, lane, acc);
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    const int64_t j = c0 + 16 * c + (lane & 15);
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int rr = 32 * rh + 16 * r + (lane >> 4) + 4 * q;
                        const int64_t cn = dg_chain_of(who, rr);
                        if ((rr >> 2) < who_m && cn < n_chains && j < D) st1(A.geval, cn * D + j, -acc[r][c][q]);
                    }
                }
        }
    }
    wait_vm0();
    if (profiling) pc2 = (long long)__builtin_readcyclecounter();
    __syncthreads();
    if (wib == 0 && lane == 0) {
        __hip_atomic_fetch_add(done, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        dg_wait(done, want, abort_word);
    }
    __syncthreads();
    if (inv_both) asm volatile("buffer_inv sc1" ::: "memory");
    if (profiling && lane == 0) {
        const long long pc3 = (long long)__builtin_readcyclecounter();
        prof[8] += pc1 - pc0; prof[9] += pc2 - pc1; prof[10] += pc3 - pc2; prof[11] += wall_clock64() - pw0; prof[12] += 1;
        prof[6] = wall_clock64();
    }
}

// ----------------------------------------------------------------------------------------
// the per-chain machine
// ----------------------------------------------------------------------------------------

// Register mirror of the cursor state: lives in locals of Machine::run() (never escapes to a
// noinline function, so it stays in VGPRs).  reg_q / reg_p name the Q-pool buffer / P-slot mirrored.
template <int NV>
struct Regs {
    double2 q[NV], g[NV], p[NV], r[NV], s[NV];
    double2 v[NV], sd[NV];   // low-rank metric (Machine<..., LR>): the cursor's velocity M^-1 p, sqrt(sigma^2)
    int64_t reg_q = -1, reg_p = -1;
    bool sig_ok = false;
    bool dirty_qg = false, dirty_pr = false;  // the registers hold the only copy (stores were elided)
    int64_t ring_leaf0 = -1, ring_leaf1 = -1;  // leaf numbers held by the LDS ring (slot 0: leaf%4==1, slot 1: leaf%4==2)
    __device__ __forceinline__ void invalidate() {
        reg_q = -1; reg_p = -1; sig_ok = false; dirty_qg = false; dirty_pr = false; ring_leaf0 = -1; ring_leaf1 = -1;
    }
};

// Streaming cache (NV < 0, -NV chunks per wave): sigma^2 and the cursor's (grad, p, rho) of this wave's chunks stay
// in VGPRs between consecutive leaves of the memory-resident kernels.  Every store still happens — memory stays
// complete, so rare paths, merge criteria and chunk-edge recomputation are untouched — the loads do not.
template <int NS>
struct SCache {
    double2 s[NS], g[NS], p[NS], r[NS];
    int64_t tag_q = -1, tag_p = -1;
    bool sig_ok = false;
    __device__ __forceinline__ void invalidate() { tag_q = -1; tag_p = -1; sig_ok = false; }
};

// NV > 0 selects the register-resident specialisation (requires FUSED; W = 1 with dim <= 128 * NV, or W = 2 / 4 waves
// per chain with ld == 128 * W * NV): the cursor state (q, grad, p, rho) and sigma^2 live in VGPRs across leapfrogs,
// so a leapfrog issues no loads at all — only the stores of the new state, which later U-turn checks / draws may read.
// LEAN (NV > 0, W >= 2, one workgroup = one chain; D up to 128 * W * NV): the register-resident design for rows too long
// for whole-vector temporaries.  The cursor (q, grad, p, rho) lives in VGPRs, sigma^2 in LDS; EVERY merge operand is
// streamed from its P-slot chunk by chunk against the resident leaf (leaf_lean), so nothing but the state itself is
// ever held as a full vector.  The same summation order as every other kernel: chunk c belongs to wave c mod W.
// REMOTE (NV > 0, W == 1, not FUSED): a host-callback model driven like a fused one — the kernel stays resident, and an
// evaluation is a remote call in the middle of the leaf: publish the position in the host's staging row, arrive, wait for
// the host's word, read (logp, gradient) back (remote_sync).  The cursor state stays in registers across the call.
// LR (NV > 0, not LEAN; round 4): the register-resident leaf under the low-rank metric M^-1 = D^1/2 (I + V (Lambda - I) V') D^1/2 —
// the cursor's velocity v = M^-1 p is a sixth resident vector, every P-slot carries it as a third vector (Args::pvec = 3), the k
// columns of V are streamed from L2 against the resident momentum (k dots + k updates per half step), and the LDS ring is not used.
// TAG: nothing but a distinct type — the out-of-line rare paths (rare_end_draw, rare_phase_fn) are members, and a function shared by two
// kernels with different register budgets (k_advance<..., WIDE>) is compiled for the larger one: the capped kernel would report its
// callee's 288 registers and lose its second wave per SIMD (measured: 4096 chains of D = 256 at 396 instead of 614 M leapfrogs/s)
template <bool FUSED, int W, int NV = 0, bool LEAN = false, bool REMOTE = false, bool LR = false, int TAG = 0>
struct Machine {
    static constexpr bool INK = FUSED || REMOTE;   // evaluations happen inside the kernel: a launch runs many steps
    // launch-per-evaluation kernels: the end of a draw is cut into slices of a launch each (engine_types.h: PH_DRAW_END / PH_DRAW_BEGIN)
    static constexpr bool SLICED = !INK && NV == 0;
    static constexpr bool DENS = REMOTE && (NPHIP_JIT != 0);   // ... by calling the model's own device function (runtime-compiled density)
    // NORING (developer builds only, -DNPHIP_DEV_W1NV=9..12): one wave per chain with more than 8 chunks per lane — the leaf of the 8-chunk kernels without
    // their LDS ring (four rings of that width do not fit a CU's LDS): every (p, rho) summary goes to its P-slot, the level-1 merges read them back from L2.
    // Measured in round 6 against the shipped two waves per chain (profiles/r6_step_at_d1025_one_wave_no_ring_rejected.txt): D = 1100 114.5 against 123.2 M
    // leapfrogs/s, D = 1280 68 against 124 (the 10-chunk leaf spills), D = 1536 44 against 110 — the ring is worth more than the second wave costs.  Not shipped.
    static constexpr bool NORING = !LEAN && !LR && W == 1 && NV > 8;
    static constexpr bool DG = REMOTE && (NPHIP_JIT == 0) && TAG == 2;   // ... by the launch-wide GEMM of the dense-precision Gaussian (dg_round)
    LdsDouble dens_lds = nullptr;   // DENS: this wave's LDS scratch for the density
    LdsDouble dens_shared = nullptr;   // DENS: the workgroup's shared LDS (staged by nphip_density_stage at kernel start)
    LdsDouble dens_rows = nullptr;     // DENS: this wave's position row [ld] and gradient row [ld] in LDS (the leaf's evaluations)
    static constexpr int NVX = NV > 0 ? NV : 1;
    static constexpr int NSX = NV < -1 ? -NV : 1;   // NV = -NS (NS >= 2): cache (sigma^2, grad, p, rho) of NS chunks per wave
    // kernel arguments, read through the constant address space: s_load into SGPRs (uniform), never flat
    const NPHIP_CONST Args& A;
    LdsCtl c;        // this wave's private LDS copy
    LdsDouble red;   // LDS reduction scratch: two areas of [16*W], used alternately (one barrier per reduction; up to 16 values each)
    LdsDouble parked = nullptr;   // callback kernels, W > 1: wave totals of the fused leaf's sums, [kParkMax][W] (leaf_cb)
    int rflip = 0;
    LdsDouble par;   // NV > 0: LDS copy of the fused model: mu[ld], a[ld], then b shifted by one (par_b[i] = b_{i-1})
    LdsDouble ring;  // NV > 0: this wave's LDS ring of two (p, rho) summaries: [slot][p|rho][NV*64 double2]
    LdsDouble edge;  // NV > 0, W > 1: chunk-edge exchange buffer of the chain [2 * chunks]
    LdsDouble sig_lds = nullptr;  // memory-resident kernels with W >= 8: LDS copy of this chain's sigma^2 (set by run())
    int64_t chain;   // local chain index
    uint32_t gchain; // global chain id (RNG key)
    int lane, wave;
    int64_t D, ld, nch;
    double* qp;      // Q-pool base of this chain
    double* pp;      // P-slot base of this chain
    double* sig2;
    double* est;
    int64_t T;
    using RegsT = Regs<NVX>;
    using SCacheT = SCache<NSX>;

    __device__ __forceinline__ Machine(const NPHIP_CONST Args& a, LdsCtl ctl, LdsDouble r, int64_t ch, LdsDouble par_ = nullptr, LdsDouble ring_ = nullptr,
                                       LdsDouble edge_ = nullptr)
        : A(a), c(ctl), red(r), par(par_), ring(ring_), edge(edge_), chain(ch), gchain((uint32_t)(a.chain_offset + ch)) {
        lane = threadIdx.x & 63;
        wave = (W == 1) ? 0 : __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
        D = a.dim; ld = a.ld; nch = ld / NPHIP_CHUNK;
        qp = a.qpool + (size_t)ch * a.nqpool * 2 * ld;
        pp = a.pslots + (size_t)ch * a.npslots * ((NV == -1 || LR) ? (size_t)a.pvec : (size_t)2) * ld;
        sig2 = a.sig2 + (size_t)ch * ld;
        est = a.est + (size_t)ch * 8 * ld;
        T = a.s.num_tune + a.s.num_draws;
        if (DENS) {   // (also in the machines the rare paths rebuild: the density is evaluated there too — initial point, step-size search)
            extern __shared__ __attribute__((aligned(16))) double s_dyn_dens[];
            // one wave per chain: four chains per workgroup, each with its own scratch and rows; several waves per chain: one chain
            const int slot = (W == 1) ? __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) : 0;
            dens_lds = (LdsDouble)s_dyn_dens + (size_t)slot * a.dens_lds_doubles;
            dens_shared = (LdsDouble)s_dyn_dens + (size_t)(W == 1 ? 4 : 1) * a.dens_lds_doubles;
            dens_rows = dens_shared + a.dens_shared_doubles + (size_t)slot * 2 * ld;
        }
    }

    // this chain's rows of the dense staging buffers (callback models)
    __device__ __forceinline__ void qe_st(int64_t i, double2 v) const { st2_dense(A.qeval + (size_t)chain * D, i, D, v); }
    // (DG: the row was written by other workgroups inside this launch — agent-scope loads, which no L1 serves: dg_round)
    __device__ __forceinline__ double2 ge_ld(int64_t i) const {
        const double* row = A.geval + (size_t)chain * D;
        if (DG && !(A.dg_variant & 128)) {
            double2 v;
            v.x = (i < D) ? __hip_atomic_load(row + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
            v.y = (i + 1 < D) ? __hip_atomic_load(row + i + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
            return v;
        }
        return ld2_dense(row, i, D);
    }
    __device__ __forceinline__ double* Q(int64_t b) const { return qp + (size_t)b * 2 * ld; }
    __device__ __forceinline__ double* G(int64_t b) const { return qp + (size_t)b * 2 * ld + ld; }
    // vectors per P-slot: (p, rho), and with the low-rank metric (memory-resident kernels only) the velocity v = M^-1 p as well
    __device__ __forceinline__ size_t pvec() const { return LRK ? (size_t)A.pvec : (size_t)2; }
    __device__ __forceinline__ double* P(int64_t s) const { return pp + (size_t)s * pvec() * ld; }
    __device__ __forceinline__ double* R(int64_t s) const { return pp + (size_t)s * pvec() * ld + ld; }
    __device__ __forceinline__ double* VEL(int64_t s) const { return pp + (size_t)s * pvec() * ld + 2 * ld; }
    // NV == -1: the memory-resident kernels WITH the low-rank metric — their own instantiations, so that the plain memory-resident
    // kernels (NV == 0: every launch-per-evaluation callback job runs each of their paths once per launch) do not carry the code
    // (measured: config 3 behind the device callback 13.7 -> 10.4 M leapfrogs/s with the low-rank branches compiled in, 29 against
    // 18 us per launch)
    static constexpr bool LRK = NV == -1 || LR;
    __device__ __forceinline__ bool lr_job() const { return LRK && A.lr_on != 0; }
    __device__ __forceinline__ const float* LRV(int j) const { return A.lr_V + ((size_t)chain * kLrMax + j) * ld; }
    __device__ __forceinline__ double* EST(int64_t e, int k) const { return est + (size_t)(e * 4 + k) * ld; }
    __device__ __forceinline__ bool leader() const { return lane == 0 && wave == 0; }
    // chain-wide sums.  Consecutive reductions alternate between two LDS areas: by the time an area is written again
    // every wave has passed the barrier of the reduction in between, i.e. has finished reading it.
    template <int N>
    __device__ __forceinline__ void rsum(double (&v)[N]) {
        static_assert(N <= 16, "a reduction area holds 16 values per wave");
        reduceN<W, N, false, !LEAN>(v, red + (W > 1 ? rflip * 16 * W : 0));
        rflip ^= 1;
    }
    __device__ __forceinline__ void rsum2(double& a, double& b) {
        double v[2] = {a, b};
        rsum(v);
        a = v[0]; b = v[1];
    }
    // sigma^2 of the hot streaming passes: from the chain's LDS copy where there is one (one chain per CU at W >= 8)
    __device__ __forceinline__ double2 sg2(int64_t i) const {
        if (sig_lds) { const double2 v = *(const NPHIP_LDS double2*)(sig_lds + i); return v; }
        return ld2(sig2, i);
    }
    __device__ __forceinline__ double sg1(int64_t i) const { return sig_lds ? sig_lds[i] : ld1(sig2, i); }

#define NPHIP_FOR_CHUNKS(i) for (int64_t cc_ = wave, i = cc_ * NPHIP_CHUNK + 2 * lane; cc_ < nch; cc_ += W, i = cc_ * NPHIP_CHUNK + 2 * lane)
    // The same walk U chunks at a time with the reads of all U issued before anything is computed or stored.  The plain loop above
    // serialises on memory: the compiler cannot move the loads of chunk k+1 above the stores of chunk k (all vectors of a chain live
    // in one pool), so a wave waits out one L2 / HBM round trip per chunk and pass — with one wave per chain at D = 1000 that was
    // 148 us per launch of the callback kernel, four times the time its traffic takes.  Chunks are visited in the same order, so
    // every (lane, component) accumulator sums in the same order: the same bits.
    template <int U, class LoadT, class BodyT>
    __device__ __forceinline__ void chunks_pf(LoadT load, BodyT body) const {
        for (int64_t c0 = wave; c0 < nch; c0 += (int64_t)W * U) {
            decltype(load((int64_t)0)) v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                int64_t cc = c0 + (int64_t)u * W;
                cc = cc < nch ? cc : nch - 1;          // (past the end: the last chunk again, never used — unconditional reads stay in registers)
                v[u] = load(cc * NPHIP_CHUNK + 2 * lane);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int64_t cc = c0 + (int64_t)u * W;
                if (cc < nch) body(cc * NPHIP_CHUNK + 2 * lane, v[u]);
            }
        }
    }
    // Used where one wave owns the whole row (W == 1: callbacks up to D = 1024, the launch-per-evaluation kernels); with several
    // waves per chain the waves of a chain overlap each other's round trips and the extra code costs more than it saves (these
    // kernels run every path once per launch: measured at D = 1000, 8 waves per chain, 84 -> 137 us per launch).
    static constexpr bool BATCHED = (W == 1);
    // ENDOUT / PFRARE — what the end of round 6 changed about the END OF A DRAW, and for which kernels (profiles/r6_call_placement_and_draw_end.txt):
    //   ENDOUT: the leaf reports how the draw ended and run() makes the one out-of-line call behind the loop of leaves (end_code, leaf_reg);
    //   PFRARE: the draw end's vector passes (gradient of the new draw, trace row + estimators, momentum) read four chunks ahead of their stores
    //           (chunks_pf<4>), and the fused gradient reads its neighbours unconditionally (tridiag_load / tridiag_eval).
    // Both for the one-wave register-resident kernels of the fused models with 2 .. 8 chunks per lane (128 < D <= 1024), chosen by same-box A/B per
    // family (warm-up, 1024 chains, M leapfrogs/s before -> after): 2 chunks 382 -> 418, 3: 292 -> 372, 4: 332 -> 343, 6: 255 -> 259, 8: 217 -> 225;
    // NOT 1 chunk (D = 10: 367 -> 320, D = 100: 457 -> 450) and NOT two waves per chain (D = 1500: 107.4 -> 101.1).  Everything else keeps its source
    // as it was, to the letter: with inter-procedural register allocation the HOT loop of a kernel is re-coloured by any change to the functions it
    // calls — with the new passes in every family the lean kernel at D = 10 000 went from 33.9 to 48.7 ms per launch (2 -> 162 spilled VGPRs in a
    // kernel not one line of which had changed) and the compiled densities lost 9 %; and the four-wave lean kernels end in a memory access fault of
    // the GPU with the position AND the momentum pass in the read-ahead form (either alone runs and is bit-identical, as are the eight-wave kernels:
    // a lead at the very end: the read-ahead passes become functions of their own, rare_end_draw then CALLS, and under inter-procedural allocation the
    // caller's view of what it clobbers may miss the nested callee — the same passes alone fault in the two-wave kernels, and run with the call behind
    // the loop or without inter-procedural allocation; not verified in the ISA).  The low-rank leaf of the same geometry (LR, W == 1) is in the
    // family: same-box D = 1000, k = 16: 14.0 -> 16.1 M leapfrogs/s (scratch/lr_reg.py; its draw end calls sample_momentum in either form).
#ifdef NPHIP_NO_PFRARE
    static constexpr bool PFRARE = false;   // (developer builds: the plain loops)
#else
    static constexpr bool PFRARE = FUSED && W == 1 && NV >= 2 && !LEAN;
#endif
    static constexpr bool ENDOUT = FUSED && W == 1 && NV >= 2 && !LEAN;
    template <class LoadT, class BodyT>
    __device__ __forceinline__ void chunks(LoadT load, BodyT body) const {
        if (nch > 2) chunks_pf<4>(load, body);
        else chunks_pf<2>(load, body);
    }
    struct V4 { double2 a, b, c, d; };

    // ------------------------------------------------------------------ scalar helpers
    __device__ void da_init(double step) {
        c->da_log_step = nphip_log(step);
        c->da_log_step_adapted = c->da_log_step;
        c->da_hbar = 0.0;
        c->da_mu = A.s.adapt_adam ? 0.0 : nphip_log(10.0 * step);
        c->adam_b1t = 1.0; c->adam_b2t = 1.0;
        c->da_count = 1;
    }
    __device__ void da_advance(double accept) {
        if (A.s.adapt_adam) {
            // Adam on log(step size) (oracle: DualAverage::advance_adam)
            const double g = accept - A.s.target_accept;
            c->da_hbar = 0.9 * c->da_hbar + (1.0 - 0.9) * g;
            c->da_mu = 0.999 * c->da_mu + (1.0 - 0.999) * (g * g);
            c->adam_b1t = c->adam_b1t * 0.9; c->adam_b2t = c->adam_b2t * 0.999;
            const double mhat = c->da_hbar / (1.0 - c->adam_b1t), vhat = c->da_mu / (1.0 - c->adam_b2t);
            c->da_log_step = c->da_log_step + A.s.adam_lr * mhat / (sqrt(vhat) + 1e-8);
            c->da_log_step_adapted = c->da_log_step;
            c->da_count += 1;
            return;
        }
        const double cnt = (double)c->da_count;
        double w = 1.0 / (cnt + A.s.da_t0);
        c->da_hbar = (1.0 - w) * c->da_hbar + w * (A.s.target_accept - accept);
        c->da_log_step = c->da_mu - c->da_hbar * sqrt(cnt) / A.s.da_gamma;
        double mk = nphip_exp(-A.s.da_k * nphip_log(cnt));
        c->da_log_step_adapted = mk * c->da_log_step + (1.0 - mk) * c->da_log_step_adapted;
        c->da_count += 1;
    }
    __device__ void update_stepsize(int64_t draw, bool best) {
        if (A.s.fixed_step_size) return;
        double step = best ? nphip_exp(c->da_log_step_adapted) : nphip_exp(c->da_log_step);
        if (A.s.jitter > 0.0) {
            nphip_u32x4 r = nphip_philox(A.s.seed, 0u, gchain, (uint32_t)draw, NPHIP_RNG_JITTER);
            double u = nphip_u01(r.v[0], r.v[1]);
            step *= fma(2.0 * A.s.jitter, u, 1.0 - A.s.jitter);
        }
        if (step > A.s.max_step_size) step = A.s.max_step_size;
        c->step_size = step;
    }
    __device__ int64_t alloc_q(bool in_tree) const {
        uint32_t used = 1u << c->cand_q;
        if (in_tree) {
            used |= (1u << c->endq[0]) | (1u << c->endq[1]) | (1u << c->curq);
            // open levels = set bits of the leaf counter (2-3 on average, not kMaxDepthCap LDS reads)
            for (uint64_t m = (uint64_t)c->nleaf & ((1ull << kMaxDepthCap) - 1); m != 0; m &= m - 1) used |= 1u << c->sub_q[__builtin_ctzll(m)];
        }
        return (int64_t)__builtin_ctz(~used);
    }
    __device__ void finish_chain(int64_t phase, int64_t err) {
        c->phase = phase;
        c->err = err;
        if (leader()) atomicAdd(&A.counters[phase == PH_ERROR ? 1 : 0], 1ull);
    }

    // ------------------------------------------------------------------ vector passes
    // Initial position (Model::init_position: src/pyfunc.rs:540-544, src/stan.rs:798-808, src/pymc.rs:505-534)
    __device__ void gen_init(int64_t attempt) {
        double* q = Q(0);
        NPHIP_FOR_CHUNKS(i) {
            double2 v = {0.0, 0.0};
            if (A.s.init_kind == 2) {
                v = ld2_dense(A.init_points + (size_t)chain * D, i, D);
            } else if (i < D) {
                nphip_u32x4 r = nphip_philox(A.s.seed, (uint32_t)(i >> 1), gchain, (uint32_t)attempt, NPHIP_RNG_INIT);
                if (A.s.init_kind == 0) {
                    v.x = fma(4.0, nphip_u01(r.v[0], r.v[1]), -2.0);
                    v.y = fma(4.0, nphip_u01(r.v[2], r.v[3]), -2.0);
                } else {
                    nphip_normal_pair(r, &v.x, &v.y);
                }
                if (i + 1 >= D) v.y = 0.0;
            }
            st2(q, i, v);
            if (!FUSED) qe_st(i, v);
        }
        c->eval_buf = 0;
    }

    // ---- low-rank metric M^-1 = D^1/2 (I + V (Lambda - I) V') D^1/2  (reference: src/wrapper.rs:307-334, the mass matrix of
    // adaptation="low_rank"; D = diag(sigma^2), V: lr_k orthonormal columns, Lambda their eigenvalues).  The contract, shared
    // with the oracle (Hamiltonian::velocity):
    //     u = std * p ;  d_j = <V_j, u> (the contract's dot) ;  c_j = (lambda_j - 1) * d_j ;
    //     w = u, then w = fma(V_j, c_j, w) for j = 0 .. k-1 ;  v = std * w
    // and for the momentum draw  p = (z + sum_j V_j f_j) * sqrt(1 / sigma^2),  f_j = (1 / sqrt(lambda_j) - 1) * <V_j, z>.
    // k dots of one vector: accumulate (pass over this wave's chunks), then two 8-value reductions.
    struct LrAcc { double2 a[kLrMax]; };
    __device__ __forceinline__ void lr_zero(LrAcc& S_) const {
#pragma unroll
        for (int j = 0; j < kLrMax; ++j) { S_.a[j].x = 0.0; S_.a[j].y = 0.0; }
    }
    // The columns are taken four at a time where four are live: a load per column behind its own `j < k` branch is a round trip
    // to L2 per column (nothing may be loaded ahead of the branch that guards it), four loads behind one branch are one round
    // trip.  Same operations on the same values in the same order per element as a column at a time.
    __device__ __forceinline__ void lr_acc(LrAcc& S_, int k, int64_t i, const double2 u) const {
        const int kq = k & ~3;
#pragma unroll
        for (int j0 = 0; j0 < kLrMax; j0 += 4) if (j0 < kq) {
            double2 vj[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) vj[t] = ld2f(LRV(j0 + t), i);
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                S_.a[j0 + t].x = fma(vj[t].x, u.x, S_.a[j0 + t].x);
                S_.a[j0 + t].y = fma(vj[t].y, u.y, S_.a[j0 + t].y);
            }
        }
        if (kq != k) {
#pragma unroll
            for (int j = 0; j < kLrMax; ++j) if (j >= kq && j < k) {
                const double2 vj = ld2f(LRV(j), i);
                S_.a[j].x = fma(vj.x, u.x, S_.a[j].x);
                S_.a[j].y = fma(vj.y, u.y, S_.a[j].y);
            }
        }
    }
    // dots -> coefficients: which = 0: c_j = (lambda_j - 1) d_j (velocity); 1: f_j = (1 / sqrt(lambda_j) - 1) d_j (momentum draw)
    __device__ __forceinline__ void lr_coef(const LrAcc& S_, int k, int which, double (&cf)[kLrMax]) {
        double v0[8], v1[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) { v0[j] = S_.a[j].x + S_.a[j].y; v1[j] = S_.a[j + 8].x + S_.a[j + 8].y; }
        rsum(v0);
        if (k > 8) rsum(v1);
#pragma unroll
        for (int j = 0; j < kLrMax; ++j) {
            const double d = j < 8 ? v0[j] : v1[j - 8];
            const double lam = j < k ? A.lr_lam[(size_t)chain * kLrMax + j] : 1.0;
            cf[j] = j < k ? (which == 0 ? (lam - 1.0) * d : (1.0 / sqrt(lam) - 1.0) * d) : 0.0;
        }
    }
    __device__ __forceinline__ double2 lr_apply(int k, int64_t i, double2 w, const double (&cf)[kLrMax]) const {
        const int kq = k & ~3;
#pragma unroll
        for (int j0 = 0; j0 < kLrMax; j0 += 4) if (j0 < kq) {
            double2 vj[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) vj[t] = ld2f(LRV(j0 + t), i);
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                w.x = fma(vj[t].x, cf[j0 + t], w.x);
                w.y = fma(vj[t].y, cf[j0 + t], w.y);
            }
        }
        if (kq != k) {
#pragma unroll
            for (int j = 0; j < kLrMax; ++j) if (j >= kq && j < k) {
                const double2 vj = ld2f(LRV(j), i);
                w.x = fma(vj.x, cf[j], w.x);
                w.y = fma(vj.y, cf[j], w.y);
            }
        }
        return w;
    }

    // The register-resident leaf takes a whole pass over the columns without a branch: the number of columns rounded up to a
    // multiple of four is a compile-time constant inside `lr_dispatch`, so every load of the pass (all chunks, all columns) can be
    // in flight at once — one round trip to L2 per pass instead of one per chunk and group.  Columns k .. KC-1 are zero in memory
    // (k_set_metric); their dots are dropped by lr_coef and their coefficients are 0.
    template <int N> struct LrCols { static constexpr int value = N; };
    template <class F> __device__ __forceinline__ void lr_dispatch(int k, F&& f) {
        switch ((k + 3) >> 2) {
            case 0: f(LrCols<0>{}); break;   // (a metric without columns: the passes still do their diagonal part)
            case 1: f(LrCols<4>{}); break;
            case 2: f(LrCols<8>{}); break;
            case 3: f(LrCols<12>{}); break;
            case 4: f(LrCols<16>{}); break;
            default: break;
        }
    }
    template <int KC> __device__ __forceinline__ void lr_acc_c(LrAcc& S_, int64_t i, const double2 u) const {
        if constexpr (KC > 0) {
            double2 vj[KC];
#pragma unroll
            for (int t = 0; t < KC; ++t) vj[t] = ld2f(LRV(t), i);
#pragma unroll
            for (int t = 0; t < KC; ++t) {
                S_.a[t].x = fma(vj[t].x, u.x, S_.a[t].x);
                S_.a[t].y = fma(vj[t].y, u.y, S_.a[t].y);
            }
        }
    }
    template <int KC> __device__ __forceinline__ double2 lr_apply_c(int64_t i, double2 w, const double (&cf)[kLrMax]) const {
        if constexpr (KC > 0) {
            double2 vj[KC];
#pragma unroll
            for (int t = 0; t < KC; ++t) vj[t] = ld2f(LRV(t), i);
#pragma unroll
            for (int t = 0; t < KC; ++t) {
                w.x = fma(vj[t].x, cf[t], w.x);
                w.y = fma(vj[t].y, cf[t], w.y);
            }
        }
        return w;
    }

    // Momentum refresh with the low-rank metric; also writes the velocity of the new state.  Returns K.
    __device__ double sample_momentum_lr(uint32_t purpose, uint32_t id) {
        double *p = P(kSlotInit), *r = R(kSlotInit), *vv = VEL(kSlotInit);
        const int k = (int)c->lr_k;
        const double* sd = A.lr_std + (size_t)chain * ld;
        LrAcc S_;
        double cf[kLrMax];
        lr_zero(S_);
        NPHIP_FOR_CHUNKS(i) {   // z (parked in the p vector) and its products with the columns
            double2 z = {0.0, 0.0};
            if (i < D) {
                nphip_normal_pair(nphip_philox(A.s.seed, (uint32_t)(i >> 1), gchain, id, purpose), &z.x, &z.y);
                if (i + 1 >= D) z.y = 0.0;
            }
            st2(p, i, z);
            lr_acc(S_, k, i, z);
        }
        lr_coef(S_, k, 1, cf);
        lr_zero(S_);
        NPHIP_FOR_CHUNKS(i) {   // p, rho = p, and the products of u = std p
            const double2 z = ld2(p, i), s2 = ld2(sig2, i), s1 = ld2(sd, i);
            double2 w = lr_apply(k, i, z, cf), pv, u;
            pv.x = (i < D) ? w.x * sqrt(1.0 / s2.x) : 0.0;
            pv.y = (i + 1 < D) ? w.y * sqrt(1.0 / s2.y) : 0.0;
            st2(p, i, pv);
            st2(r, i, pv);
            u.x = s1.x * pv.x; u.y = s1.y * pv.y;
            lr_acc(S_, k, i, u);
        }
        lr_coef(S_, k, 0, cf);
        double2 acc = {0.0, 0.0};
        NPHIP_FOR_CHUNKS(i) {   // v = std (u + V c), K
            const double2 pv = ld2(p, i), s1 = ld2(sd, i);
            double2 u, v;
            u.x = s1.x * pv.x; u.y = s1.y * pv.y;
            const double2 w = lr_apply(k, i, u, cf);
            v.x = s1.x * w.x; v.y = s1.y * w.y;
            st2(vv, i, v);
            acc.x = fma(pv.x, v.x, acc.x);
            acc.y = fma(pv.y, v.y, acc.y);
        }
        double a = acc.x + acc.y, b = 0.0;
        rsum2(a, b);
        return 0.5 * a;
    }

    // Momentum refresh (initialize_trajectory): p = z * sqrt(1/sig2); rho = p.  Returns K.
    __device__ double sample_momentum(uint32_t purpose, uint32_t id) {
        if (lr_job() && c->host_metric) return sample_momentum_lr(purpose, id);
        double* p = P(kSlotInit);
        double* r = R(kSlotInit);
        double2 acc = {0.0, 0.0};
        if (PFRARE) {   // sigma^2 of four chunks read ahead — a read behind the stores of the chunk before waits for them to complete
            chunks_pf<4>([&](int64_t i) { return ld2(sig2, i); },
                         [&](int64_t i, const double2& s2) {
                    double2 v = {0.0, 0.0};
                    if (i < D) {
                        double z0, z1;
                        nphip_normal_pair(nphip_philox(A.s.seed, (uint32_t)(i >> 1), gchain, id, purpose), &z0, &z1);
                        v.x = z0 * sqrt(1.0 / s2.x);
                        if (i + 1 < D) v.y = z1 * sqrt(1.0 / s2.y);
                    }
                    st2(p, i, v);
                    st2(r, i, v);
                    if (lr_job()) { double2 vel; vel.x = s2.x * v.x; vel.y = s2.y * v.y; st2(VEL(kSlotInit), i, vel); }
                    acc.x = fma(v.x, s2.x * v.x, acc.x);
                    acc.y = fma(v.y, s2.y * v.y, acc.y);
                         });
        } else
        NPHIP_FOR_CHUNKS(i) {
            double2 s2 = ld2(sig2, i);
            double2 v = {0.0, 0.0};
            if (i < D) {
                double z0, z1;
                nphip_normal_pair(nphip_philox(A.s.seed, (uint32_t)(i >> 1), gchain, id, purpose), &z0, &z1);
                v.x = z0 * sqrt(1.0 / s2.x);
                if (i + 1 < D) v.y = z1 * sqrt(1.0 / s2.y);
            }
            st2(p, i, v);
            st2(r, i, v);
            if (lr_job()) { double2 vel; vel.x = s2.x * v.x; vel.y = s2.y * v.y; st2(VEL(kSlotInit), i, vel); }
            acc.x = fma(v.x, s2.x * v.x, acc.x);
            acc.y = fma(v.y, s2.y * v.y, acc.y);
        }
        double a = acc.x + acc.y, b = 0.0;
        rsum2(a, b);
        return 0.5 * a;
    }

    // Leapfrog, first half: p_half = p + eps/2 g ; q' = q + eps sig2 p_half   
    __device__ __forceinline__ void lf1(int64_t srcq, int64_t srcp, int64_t newq, int64_t newp, int64_t sign, bool defer = false) {
        c->lf_srcq = srcq; c->lf_srcp = srcp; c->lf_newq = newq; c->lf_newp = newp; c->lf_sign = sign;
        c->eval_buf = newq;
        if (INK && defer) return;  // fused / resident models integrate the whole step in leaf_reg() / lf_stream()
        const double eps = (double)sign * c->step_size;
        const double h = 0.5 * eps;
        const double *q = Q(srcq), *g = G(srcq), *p = P(srcp);
        double *qn = Q(newq), *pn = P(newp);
        if (lr_job() && c->host_metric) {
            // low-rank metric: q' = q + eps v(p_half) — the half-kicked momentum and its products with the columns, then the drift
            const int k = (int)c->lr_k;
            const double* sd = A.lr_std + (size_t)chain * ld;
            LrAcc S_;
            double cf[kLrMax];
            lr_zero(S_);
            NPHIP_FOR_CHUNKS(i) {
                const double2 g2 = ld2(g, i), p2 = ld2(p, i), s1 = ld2(sd, i);
                double2 ph, u;
                ph.x = fma(h, g2.x, p2.x);
                ph.y = fma(h, g2.y, p2.y);
                st2(pn, i, ph);
                u.x = s1.x * ph.x; u.y = s1.y * ph.y;
                lr_acc(S_, k, i, u);
            }
            lr_coef(S_, k, 0, cf);
            NPHIP_FOR_CHUNKS(i) {
                const double2 q2 = ld2(q, i), ph = ld2(pn, i), s1 = ld2(sd, i);
                double2 u, qq;
                u.x = s1.x * ph.x; u.y = s1.y * ph.y;
                const double2 w = lr_apply(k, i, u, cf);
                qq.x = fma(eps, s1.x * w.x, q2.x);
                qq.y = fma(eps, s1.y * w.y, q2.y);
                st2(qn, i, qq);
                if (!FUSED) qe_st(i, qq);
            }
            if (FUSED) chain_sync<W>();
            return;
        }
        if (!BATCHED) {
            NPHIP_FOR_CHUNKS(i) {
                double2 q2 = ld2(q, i), g2 = ld2(g, i), p2 = ld2(p, i), s2 = ld2(sig2, i);
                double2 ph, qq;
                ph.x = fma(h, g2.x, p2.x);
                ph.y = fma(h, g2.y, p2.y);
                qq.x = fma(eps, s2.x * ph.x, q2.x);
                qq.y = fma(eps, s2.y * ph.y, q2.y);
                st2(qn, i, qq);
                st2(pn, i, ph);
                if (!FUSED) qe_st(i, qq);
            }
        } else
        chunks(
            [&](int64_t i) { V4 v; v.a = ld2(q, i); v.b = ld2(g, i); v.c = ld2(p, i); v.d = ld2(sig2, i); return v; },
            [&](int64_t i, const V4& v) {
                double2 ph, qq;
                ph.x = fma(h, v.b.x, v.c.x);
                ph.y = fma(h, v.b.y, v.c.y);
                qq.x = fma(eps, v.d.x * ph.x, v.a.x);
                qq.y = fma(eps, v.d.y * ph.y, v.a.y);
                st2(qn, i, qq);
                st2(pn, i, ph);
                if (!FUSED) qe_st(i, qq);
            });
        if (FUSED) chain_sync<W>();  // the fused model reads neighbouring elements of q'
    }

    // Fused tridiagonal-Gaussian gradient for the pair at i (nphip model contract).
    __device__ __forceinline__ void tridiag_pair(const double* q, int64_t i, double2& z, double2& g) const {
        double2 q2 = ld2(q, i), mu = ld2(A.m_mu, i), a = ld2(A.m_a, i), b = ld2(A.m_b, i);
        z.x = q2.x - mu.x;
        z.y = q2.y - mu.y;
        double tx = a.x * z.x;
        if (i > 0) tx = fma(ld1(A.m_b, i - 1), ld1(q, i - 1) - ld1(A.m_mu, i - 1), tx);
        if (i + 1 < D) tx = fma(b.x, z.y, tx);
        double ty = a.y * z.y;
        ty = fma(b.x, z.x, ty);
        if (i + 2 < D) ty = fma(b.y, ld1(q, i + 2) - ld1(A.m_mu, i + 2), ty);
        g.x = -tx;
        g.y = (i + 1 < D) ? -ty : 0.0;
        if (i >= D) g.x = 0.0;
    }
    // Fused tridiagonal-Gaussian gradient for the pair at i — the form of the ENDOUT families: what it reads, and what it computes from that.
    // Every read is unconditional (the neighbours' indices clamped, their terms taken or not by a select: the same operations on the same values) — a
    // read behind a lane's condition is a block of its own with its own wait: three round trips to memory per chunk instead of one (round 6: 32 k
    // cycles per draw for the gradient of the new draw at D = 1000).  Apart, so that a pass can have the reads of several chunks in flight (PFRARE).
    struct TriIn { double2 q, mu, a, b; double bm, qm, mum, qp, mup; };
    __device__ __forceinline__ TriIn tridiag_load(const double* q, int64_t i) const {
        TriIn v;
        const int64_t im = i > 0 ? i - 1 : 0, ip = i + 2 < D ? i + 2 : i;
        v.q = ld2(q, i); v.mu = ld2(A.m_mu, i); v.a = ld2(A.m_a, i); v.b = ld2(A.m_b, i);
        v.bm = ld1(A.m_b, im); v.qm = ld1(q, im); v.mum = ld1(A.m_mu, im);
        v.qp = ld1(q, ip); v.mup = ld1(A.m_mu, ip);
        return v;
    }
    __device__ __forceinline__ void tridiag_eval(const TriIn& v, int64_t i, double2& z, double2& g) const {
        z.x = v.q.x - v.mu.x;
        z.y = v.q.y - v.mu.y;
        double tx = v.a.x * z.x;
        const double tx1 = fma(v.bm, v.qm - v.mum, tx);
        tx = (i > 0) ? tx1 : tx;
        const double tx2 = fma(v.b.x, z.y, tx);
        tx = (i + 1 < D) ? tx2 : tx;
        double ty = v.a.y * z.y;
        ty = fma(v.b.x, z.x, ty);
        const double ty1 = fma(v.b.y, v.qp - v.mup, ty);
        ty = (i + 2 < D) ? ty1 : ty;
        g.x = -tx;
        g.y = (i + 1 < D) ? -ty : 0.0;
        if (i >= D) g.x = 0.0;
    }

    // Position-only evaluation (initial point).  FUSED: compute; callback: copy staged gradient.
    __device__ void eval_position(int64_t buf, double& lp, int64_t& code) {
        double* g = G(buf);
        if (FUSED) {
            const double* q = Q(buf);
            double2 acc = {0.0, 0.0};
            if (PFRARE) {
                chunks_pf<4>([&](int64_t i) { return tridiag_load(q, i); },
                             [&](int64_t i, const TriIn& v) {
                                 double2 z, gg;
                                 tridiag_eval(v, i, z, gg);
                                 st2(g, i, gg);
                                 acc.x = fma(z.x, gg.x, acc.x);
                                 acc.y = fma(z.y, gg.y, acc.y);
                             });
            } else
            NPHIP_FOR_CHUNKS(i) {
                double2 z, gg;
                tridiag_pair(q, i, z, gg);
                st2(g, i, gg);
                acc.x = fma(z.x, gg.x, acc.x);
                acc.y = fma(z.y, gg.y, acc.y);
            }
            double a = acc.x + acc.y, b = 0.0;
            rsum2(a, b);
            lp = 0.5 * a;
            code = 0;
        } else {
            if (REMOTE) remote_eval(lp, code);
            else { lp = A.ueval[chain]; code = A.ecode ? A.ecode[chain] : 0; }
            NPHIP_FOR_CHUNKS(i) st2(g, i, ge_ld(i));
        }
    }

    // ---- resident host-callback launches: one evaluation = one rendezvous of the group with the host.
    // Every chain of the group arrives once (its position is in the host's staging row and released to the system first); the
    // last arriver publishes the job-wide done / error counts and the evaluation's sequence number in pinned host memory,
    // polls the host's word over PCIe and republishes it in device memory for the other chains.  The host answers with the
    // sequence number (results are in the staging rows), optionally | kGoLast: finish this step, then leave the kernel at the
    // next boundary.  Finished chains keep taking part (run()), so the arrival count is always the group's size.  All chains
    // of the group are resident: the launch's roll call (k_advance) made sure before anyone got here.
    // The four chains of a workgroup rendezvous together: each waits for its own stores (they are in this XCD's L2 then), one
    // workgroup barrier, and wave 0 alone pays for __threadfence_system() — its release half writes back the WHOLE L2
    // (buffer_wbl2), about a microsecond per wave that issues it on the same XCD, and with every chain doing it it was a third of a
    // step.  Measured and rejected (profiles/r2_config4_resident_launches.txt): system-scope (sc0 sc1) stores or loads for the
    // staging rows instead of the fence — they go over PCIe one at a time, ~2.3 us per chain; agent-scope (sc1) stores — correct
    // but slower than the one fence per workgroup; plain stores and only a vmcnt wait — the flag overtakes the data and the
    // trace is wrong.
    __device__ __forceinline__ void remote_sync() {
        const int grp = (int)c->hs_grp;
        const unsigned seq = (unsigned)c->hs_seq;
        // (the word wave 0 hands to the other waves of the workgroup lives in wave 0's control block)
        NPHIP_LDS unsigned long long* box = (NPHIP_LDS unsigned long long*)&(c - (int)(threadIdx.x >> 6))->hs_box;
        wait_vm0();
        __syncthreads();
        if (__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) == 0) {
            unsigned long long go = 0;
            __threadfence_system();   // the positions of this workgroup's chains are in the host's staging rows
            if (lane == 0) {
                const unsigned cnt = (unsigned)c->hs_wgn;
                const unsigned old = __hip_atomic_fetch_add(&A.grp_arrive[grp], cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (old + cnt == (unsigned)c->hs_n) {
                    __hip_atomic_store(&A.grp_arrive[grp], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    // one word: the sequence number, "every chain of the job is done", "some chain is in error" (no second fence)
                    const unsigned long long done = __hip_atomic_load(&A.counters[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    const unsigned long long errs = __hip_atomic_load(&A.counters[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    volatile unsigned long long* f = A.grp_flag + 4 * grp;
                    f[0] = (unsigned long long)seq | (done >= (unsigned long long)A.n_chains ? kPubAllDone : 0ull) | (errs > 0 ? kPubError : 0ull);
                    for (;;) {
                        go = __hip_atomic_load((unsigned long long*)(A.grp_go + 8 * grp), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                        if ((go & kGoSeqMask) == (unsigned long long)seq) break;
                        __builtin_amdgcn_s_sleep(4);
                    }
                    __hip_atomic_store(&A.grp_go_dev[16 * grp], go, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                } else {
                    for (;;) {
                        go = __hip_atomic_load(&A.grp_go_dev[16 * grp], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if ((go & kGoSeqMask) == (unsigned long long)seq) break;
                        __builtin_amdgcn_s_sleep(8);
                    }
                }
                *box = go;
            }
        }
        __syncthreads();
        const unsigned long long go = *box;
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");   // system scope: no stale line of the staging rows survives
        c->hs_seq = (int64_t)(seq + 1u);
        if (go & kGoLast) c->hs_last = 1;
    }

    // One evaluation of a REMOTE machine at the position in this chain's staging row: (logp, code), gradient in the gradient row.
    // Host callbacks: the rendezvous with the host.  Runtime-compiled densities: a call of the model's device function — the
    // wave's own stores of the position are made visible to all of its lanes first, the density's stores of the gradient after.
    // (DENS, rows_in_lds: the leaf keeps the position and gradient rows of its evaluations in LDS — the density's accesses to
    //  them are LDS accesses instead of L2 round trips; the rare paths use the staging rows in memory, as the callbacks do)
    // (DG, want_lp = false: the register-resident leaf takes logp = 1/2 (x - mu) . grad out of its own pass over the gradient row — the same
    //  per-lane fma chains, summed with the leaf's other reductions: the same bits, one pass over the row and one wave reduction fewer)
    __device__ __forceinline__ void remote_eval(double& lp, int64_t& code, bool rows_in_lds = false, bool want_lp = true) {
#if NPHIP_JIT
        if (DENS) {
            chain_sync<W>();   // (several waves per chain: every wave has written its chunks of the position)
            const double* xr = rows_in_lds ? (const double*)dens_rows : A.qeval + (size_t)chain * D;
            double* gr = rows_in_lds ? (double*)(dens_rows + ld) : A.geval + (size_t)chain * D;
            // `lane` of the density = the thread's index within the chain (0 .. 64 W - 1)
            lp = nphip_density(*(const NphipData*)A.dens_data, (int)D, xr, gr, (double*)dens_lds, (const double*)dens_shared, (W == 1) ? lane : (int)threadIdx.x);
            chain_sync<W>();
            code = 0;
            return;
        }
#endif
        if (DG) {
            // the position is in this chain's staging row; one round of the launch-wide GEMM later the gradient is in its gradient row.
            // Ctl::hs_seq counts the rounds this wave has taken part in (every wave takes part in all hs_n of the launch: k_advance)
            dg_round(A, (int)c->hs_seq, A.n_chains, (NPHIP_LDS int64_t*)c->prof);
            c->hs_seq = c->hs_seq + 1;
            code = 0;
            if (!want_lp) return;
            // logp = 1/2 (x - mu) . grad in the contract's summation order (nphip_spec.h), as the launch-per-evaluation form computes it
            const double* xr = A.qeval + (size_t)chain * D;
            double2 acc = {0.0, 0.0};
            NPHIP_FOR_CHUNKS(i) {
                const double2 xv = ld2_dense(xr, i, D), gv = ge_ld(i);
                const double zx = xv.x - ((i < D) ? ld1(A.dg_mu, i) : 0.0), zy = xv.y - ((i + 1 < D) ? ld1(A.dg_mu, i + 1) : 0.0);
                acc.x = fma(zx, gv.x, acc.x);
                acc.y = fma(zy, gv.y, acc.y);
            }
            double a_ = acc.x + acc.y, b_ = 0.0;
            rsum2(a_, b_);
            lp = 0.5 * a_;
            code = 0;
            return;
        }
        remote_sync();
        lp = A.ueval[chain];
        code = A.ecode ? A.ecode[chain] : 0;
    }

    // ---- criteria of a sub-tree merge INSIDE a doubling (all leaves on one side of the origin): for an
    // (earlier, later) pair  span = (rho_late - rho_early) + p_early  in both directions (SURVEY A.4 modes 0/2).
    __device__ __forceinline__ void span_acc(double pe_, double re_, double pl, double rl, double s2v, double& a1, double& a2) const {
        const double t = (rl - re_) + pe_;
        a1 = fma(t, s2v * pl, a1);
        a2 = fma(t, s2v * pe_, a2);
    }

    // the same with the velocities v = M^-1 p of the two states given (low-rank metric: not sigma^2 p)
    __device__ __forceinline__ void span_acc_v(double pe_, double re_, double ve_, double rl, double vl, double& a1, double& a2) const {
        const double t = (rl - re_) + pe_;
        a1 = fma(t, vl, a1);
        a2 = fma(t, ve_, a2);
    }

    // ---- Streaming fused leapfrog (FUSED, NV == 0: any D, any W).  ONE pass per leaf: each wave streams its
    // chunks once — read q, grad, p, rho, sigma^2; write q', grad', p', rho' — interior tridiagonal neighbours come
    // from DPP wave shifts, the two chunk-edge neighbours are recomputed from their own inputs (bit-identical, so
    // no cross-wave exchange and no second pass), and the level-0 U-turn criterion is accumulated on the way.
    // z' = q' - mu of ONE element e (a chunk-edge neighbour owned by another lane / wave), from that element's own inputs.
    // The cached streaming kernel keeps no gradient in memory inside a tree (NOG): it is rebuilt from q (bit-identical to
    // the value computed when q was produced: same operations in the same order).
    static constexpr bool NOG = FUSED && NV < -1;   // measured: pays with the VGPR cache (W = 1), costs 4-8 % at W = 8
    __device__ __forceinline__ double elem_grad(const double* q, int64_t e) const {
        const double ze = ld1(q, e) - ld1(A.m_mu, e);
        double t = ld1(A.m_a, e) * ze;
        t = fma((e > 0) ? ld1(A.m_b, e - 1) : -0.0, (e > 0) ? ld1(q, e - 1) - ld1(A.m_mu, e - 1) : 0.0, t);
        t = fma(ld1(A.m_b, e), (e + 1 < ld) ? ld1(q, e + 1) - ld1(A.m_mu, e + 1) : 0.0, t);
        return -t;
    }
    __device__ __forceinline__ double edge_z(const double* q, const double* g, const double* p, int64_t e, double eps, double h) const {
        const double ge = NOG ? elem_grad(q, e) : ld1(g, e);
        const double ph = fma(h, ge, ld1(p, e));
        return fma(eps, sg1(e) * ph, ld1(q, e)) - ld1(A.m_mu, e);
    }
    // gradient of the element pair i of this lane from q (the chunk's other pairs by DPP, its neighbours by uniform loads)
    __device__ __forceinline__ double2 pair_grad(const double* q, int64_t i, const double2 q2) const {
        const double2 mu = ld2(A.m_mu, i), a = ld2(A.m_a, i), b = ld2(A.m_b, i);
        const int64_t c0 = i - 2 * lane;
        double2 z;
        z.x = q2.x - mu.x;
        z.y = q2.y - mu.y;
        const double zl_edge = (c0 > 0) ? ld1(q, c0 - 1) - ld1(A.m_mu, c0 - 1) : 0.0;
        const double zr_edge = (c0 + NPHIP_CHUNK < ld) ? ld1(q, c0 + NPHIP_CHUNK) - ld1(A.m_mu, c0 + NPHIP_CHUNK) : 0.0;
        const double bl = wave_shr1(b.y, (c0 > 0) ? ld1(A.m_b, c0 - 1) : -0.0);
        const double zl = wave_shr1(z.y, zl_edge);
        const double zr = wave_shl1(z.x, zr_edge);
        double tx = a.x * z.x;
        tx = fma(bl, zl, tx);
        tx = fma(b.x, z.y, tx);
        double ty = a.y * z.y;
        ty = fma(b.x, z.x, ty);
        ty = fma(b.y, zr, ty);
        double2 gq;
        gq.x = -tx;
        gq.y = -ty;
        return gq;
    }
    __device__ __forceinline__ double lf_stream(double& lp, int64_t idx_new, bool& turn0, SCacheT& Y) {
        const int64_t srcq = c->lf_srcq, srcp = c->lf_srcp, newq = c->lf_newq, newp = c->lf_newp;
        const double eps = (double)c->lf_sign * c->step_size;
        const double h = 0.5 * eps;
        const bool copy_rho = (idx_new == -1);
        const double *q = Q(srcq), *g = G(srcq), *p = P(srcp), *r = R(srcp);
        double *qn = Q(newq), *gn = G(newq), *pn = P(newp), *rn = R(newp);
        double2 accK = {0.0, 0.0}, accL = {0.0, 0.0}, accE = {0.0, 0.0}, accS = {0.0, 0.0};
        if (NV < -1) {
            // fill the cache where the cursor moved (new doubling, rare path) — normally nothing to do
            if (!Y.sig_ok) {
#pragma unroll
                for (int k = 0; k < NSX; ++k) { const int64_t cc = wave + (int64_t)k * W; if (cc < nch) Y.s[k] = ld2(sig2, cc * NPHIP_CHUNK + 2 * lane); }
                Y.sig_ok = true;
            }
            if (Y.tag_q != srcq) {
#pragma unroll
                for (int k = 0; k < NSX; ++k) {
                    const int64_t cc = wave + (int64_t)k * W;
                    if (cc < nch) { const int64_t i = cc * NPHIP_CHUNK + 2 * lane; Y.g[k] = pair_grad(q, i, ld2(q, i)); }
                }
            }
            if (Y.tag_p != srcp) {
#pragma unroll
                for (int k = 0; k < NSX; ++k) {
                    const int64_t cc = wave + (int64_t)k * W;
                    if (cc < nch) { Y.p[k] = ld2(p, cc * NPHIP_CHUNK + 2 * lane); Y.r[k] = ld2(r, cc * NPHIP_CHUNK + 2 * lane); }
                }
            }
        }
        auto body = [&](const int64_t i, const double2 q2, const double2 g2, const double2 p2, const double2 r2, const double2 s2,
                        double2& gg, double2& pv, double2& rr) {
            const double2 mu = ld2(A.m_mu, i), a = ld2(A.m_a, i), b = ld2(A.m_b, i);   // b pads are -0.0 (host)
            double2 ph, qq, z;
            ph.x = fma(h, g2.x, p2.x);
            ph.y = fma(h, g2.y, p2.y);
            qq.x = fma(eps, s2.x * ph.x, q2.x);
            qq.y = fma(eps, s2.y * ph.y, q2.y);
            z.x = qq.x - mu.x;
            z.y = qq.y - mu.y;
            // chunk-edge neighbours (uniform addresses: one broadcast transaction each)
            const int64_t c0 = i - 2 * lane;                 // first element of this chunk
            const double zl_edge = (c0 > 0) ? edge_z(q, g, p, c0 - 1, eps, h) : 0.0;
            const double zr_edge = (c0 + NPHIP_CHUNK < ld) ? edge_z(q, g, p, c0 + NPHIP_CHUNK, eps, h) : 0.0;
            const double bl = wave_shr1(b.y, (c0 > 0) ? ld1(A.m_b, c0 - 1) : -0.0);
            const double zl = wave_shr1(z.y, zl_edge);
            const double zr = wave_shl1(z.x, zr_edge);
            double tx = a.x * z.x;
            tx = fma(bl, zl, tx);
            tx = fma(b.x, z.y, tx);
            double ty = a.y * z.y;
            ty = fma(b.x, z.x, ty);
            ty = fma(b.y, zr, ty);
            gg.x = -tx;
            gg.y = -ty;
            accL.x = fma(z.x, gg.x, accL.x);
            accL.y = fma(z.y, gg.y, accL.y);
            pv.x = fma(h, gg.x, ph.x);
            pv.y = fma(h, gg.y, ph.y);
            const double vx = s2.x * pv.x, vy = s2.y * pv.y;
            accK.x = fma(pv.x, vx, accK.x);
            accK.y = fma(pv.y, vy, accK.y);
            rr.x = (copy_rho ? -0.0 : r2.x) + pv.x;
            rr.y = (copy_rho ? -0.0 : r2.y) + pv.y;
            const double tx0 = (rr.x - r2.x) + p2.x, ty0 = (rr.y - r2.y) + p2.y;
            accE.x = fma(tx0, vx, accE.x);
            accE.y = fma(ty0, vy, accE.y);
            accS.x = fma(tx0, s2.x * p2.x, accS.x);
            accS.y = fma(ty0, s2.y * p2.y, accS.y);
            st2(qn, i, qq); st2(pn, i, pv); st2(rn, i, rr);
            if (!NOG) st2(gn, i, gg);
        };
        if (NV < -1) {
            // the only state loads left: q of every chunk, issued back to back so
            // that they are all in flight together (chunks past the end re-read the last one; never used)
            double2 qv[NSX];
#pragma unroll
            for (int k = 0; k < NSX; ++k) {
                int64_t cc = wave + (int64_t)k * W;
                cc = cc < nch ? cc : nch - 1;
                qv[k] = ld2(q, cc * NPHIP_CHUNK + 2 * lane);
            }
#pragma unroll
            for (int k = 0; k < NSX; ++k) {
                const int64_t cc = wave + (int64_t)k * W;
                if (cc < nch) {
                    double2 gg, pv, rr;
                    body(cc * NPHIP_CHUNK + 2 * lane, qv[k], Y.g[k], Y.p[k], Y.r[k], Y.s[k], gg, pv, rr);
                    Y.g[k] = gg;
                    Y.p[k] = pv; Y.r[k] = rr;
                }
            }
            Y.tag_q = newq;
            Y.tag_p = newp;
        } else {
            NPHIP_FOR_CHUNKS(i) {
                double2 gg, pv, rr;
                const double2 q2 = ld2(q, i);
                body(i, q2, NOG ? pair_grad(q, i, q2) : ld2(g, i), ld2(p, i), ld2(r, i), sg2(i), gg, pv, rr);
            }
        }
        double v[4] = {accK.x + accK.y, accL.x + accL.y, accE.x + accE.y, accS.x + accS.y};
        rsum(v);
        lp = 0.5 * v[1];
        turn0 = (v[2] < 0.0) || (v[3] < 0.0);
        return 0.5 * v[0];
    }
    // The three criteria of a sub-tree merge inside a doubling in ONE streaming pass (direction-free span):
    // (A.first, TL) || (A.last, TL) || (A.first, TF)
    __device__ __forceinline__ bool check3_stream(int64_t sA, int64_t sB, int64_t sTF, int64_t sTL, SCacheT& Y) {
        const double *pa = P(sA), *ra = R(sA), *pb = P(sB), *rb = R(sB), *pf = P(sTF), *rf = R(sTF), *pl = P(sTL), *rl = R(sTL);
        double2 acc[6];
#pragma unroll
        for (int n = 0; n < 6; ++n) { acc[n].x = 0.0; acc[n].y = 0.0; }
        auto body = [&](const double2 xa, const double2 xra, const double2 xb, const double2 xrb, const double2 xf, const double2 xrf,
                        const double2 xl, const double2 xrl, const double2 s2) {
            span_acc(xa.x, xra.x, xl.x, xrl.x, s2.x, acc[0].x, acc[1].x);
            span_acc(xa.y, xra.y, xl.y, xrl.y, s2.y, acc[0].y, acc[1].y);
            span_acc(xb.x, xrb.x, xl.x, xrl.x, s2.x, acc[2].x, acc[3].x);
            span_acc(xb.y, xrb.y, xl.y, xrl.y, s2.y, acc[2].y, acc[3].y);
            span_acc(xa.x, xra.x, xf.x, xrf.x, s2.x, acc[4].x, acc[5].x);
            span_acc(xa.y, xra.y, xf.y, xrf.y, s2.y, acc[4].y, acc[5].y);
        };
        if (NV < -1 && Y.tag_p == sTL && Y.sig_ok) {   // T.last is the leaf just integrated: still in the cache
            // operands of two chunks (12 vectors) in flight at a time
#pragma unroll
            for (int k0 = 0; k0 < NSX; k0 += 2) {
                double2 o[2][6];
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    int64_t cc = wave + (int64_t)(k0 + u) * W;
                    cc = cc < nch ? cc : nch - 1;
                    const int64_t i = cc * NPHIP_CHUNK + 2 * lane;
                    o[u][0] = ld2(pa, i); o[u][1] = ld2(ra, i); o[u][2] = ld2(pb, i); o[u][3] = ld2(rb, i); o[u][4] = ld2(pf, i); o[u][5] = ld2(rf, i);
                }
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int k = k0 + u;
                    if (k < NSX && wave + (int64_t)k * W < nch) body(o[u][0], o[u][1], o[u][2], o[u][3], o[u][4], o[u][5], Y.p[k < NSX ? k : 0], Y.r[k < NSX ? k : 0], Y.s[k < NSX ? k : 0]);
                }
            }
        } else {
            NPHIP_FOR_CHUNKS(i) { body(ld2(pa, i), ld2(ra, i), ld2(pb, i), ld2(rb, i), ld2(pf, i), ld2(rf, i), ld2(pl, i), ld2(rl, i), sg2(i)); }
        }
        double v[6];
#pragma unroll
        for (int n = 0; n < 6; ++n) v[n] = acc[n].x + acc[n].y;
        rsum(v);
        return (v[0] < 0.0) || (v[1] < 0.0) || (v[2] < 0.0) || (v[3] < 0.0) || (v[4] < 0.0) || (v[5] < 0.0);
    }

    // Leapfrog, second half (+ fused gradient): p' = p_half + eps/2 g' ; K' ; rho' = rho + p'.
    // Returns kinetic energy; lp/code describe the logp evaluation.
    __device__ double lf2(double& lp, int64_t& code, int64_t idx_new) {
        const int64_t newq = c->lf_newq, newp = c->lf_newp;
        const double h = 0.5 * (double)c->lf_sign * c->step_size;
        double *g = G(newq), *pn = P(newp), *rn = R(newp);
        const double* rp = R(c->lf_srcp);
        const bool copy_rho = (idx_new == -1);
        double2 accK = {0.0, 0.0}, accL = {0.0, 0.0};
        if (!FUSED) {
            if (REMOTE) remote_eval(lp, code);
            else { lp = A.ueval[chain]; code = A.ecode ? A.ecode[chain] : 0; }
            if (code != 0 || !isfinite(lp)) return 0.0;
        }
        const double* q = Q(newq);
        const bool lrm = lr_job() && c->host_metric;
        const int lrk = lrm ? (int)c->lr_k : 0;
        const double* sd = lrm ? A.lr_std + (size_t)chain * ld : nullptr;
        LrAcc S_;
        if (lrm) lr_zero(S_);
        if (BATCHED && !FUSED && !lr_job()) {
            const double* ge = A.geval + (size_t)chain * D;
            chunks(
                [&](int64_t i) { V4 v; v.a = ld2_dense(ge, i, D); v.b = ld2(pn, i); v.c = ld2(sig2, i); v.d = ld2(rp, i); return v; },
                [&](int64_t i, const V4& v) {
                    double2 pv, rr;
                    pv.x = fma(h, v.a.x, v.b.x);
                    pv.y = fma(h, v.a.y, v.b.y);
                    const double vx = v.c.x * pv.x, vy = v.c.y * pv.y;
                    accK.x = fma(pv.x, vx, accK.x);
                    accK.y = fma(pv.y, vy, accK.y);
                    rr.x = copy_rho ? pv.x : v.d.x + pv.x;
                    rr.y = copy_rho ? pv.y : v.d.y + pv.y;
                    st2(g, i, v.a);
                    st2(pn, i, pv);
                    st2(rn, i, rr);
                });
        } else
        NPHIP_FOR_CHUNKS(i) {
            double2 gg;
            if (FUSED) {
                double2 z;
                tridiag_pair(q, i, z, gg);
                accL.x = fma(z.x, gg.x, accL.x);
                accL.y = fma(z.y, gg.y, accL.y);
            } else {
                gg = ge_ld(i);
            }
            double2 ph = ld2(pn, i), s2 = ld2(sig2, i), r2 = ld2(rp, i);
            double2 pv, rr;
            pv.x = fma(h, gg.x, ph.x);
            pv.y = fma(h, gg.y, ph.y);
            if (lrm) {   // the kinetic energy needs the velocity: products with the columns now, the rest in a second pass
                const double2 s1 = ld2(sd, i);
                double2 u;
                u.x = s1.x * pv.x; u.y = s1.y * pv.y;
                lr_acc(S_, lrk, i, u);
            } else {
                const double vx = s2.x * pv.x, vy = s2.y * pv.y;
                accK.x = fma(pv.x, vx, accK.x);
                accK.y = fma(pv.y, vy, accK.y);
                if (lr_job()) { double2 vel; vel.x = vx; vel.y = vy; st2(VEL(newp), i, vel); }
            }
            rr.x = copy_rho ? pv.x : r2.x + pv.x;
            rr.y = copy_rho ? pv.y : r2.y + pv.y;
            st2(g, i, gg);
            st2(pn, i, pv);
            st2(rn, i, rr);
        }
        if (lrm) {
            double cf[kLrMax];
            lr_coef(S_, lrk, 0, cf);
            double* vn = VEL(newp);
            NPHIP_FOR_CHUNKS(i) {
                const double2 pv = ld2(pn, i), s1 = ld2(sd, i);
                double2 u, v;
                u.x = s1.x * pv.x; u.y = s1.y * pv.y;
                const double2 w = lr_apply(lrk, i, u, cf);
                v.x = s1.x * w.x; v.y = s1.y * w.y;
                st2(vn, i, v);
                accK.x = fma(pv.x, v.x, accK.x);
                accK.y = fma(pv.y, v.y, accK.y);
            }
        }
        double a = accK.x + accK.y, b = accL.x + accL.y;
        rsum2(a, b);
        if (FUSED) { lp = 0.5 * b; code = 0; }
        return 0.5 * a;
    }


    // ======================================================================================
    // Launch-per-evaluation kernels (callback models: NV == 0, the evaluation happens between two launches): the hot leaf.
    //
    // A launch of these kernels serves ONE evaluation and lasts as long as the chain with the longest dependent path — and
    // among a thousand chains there always is one that closes a deep sub-tree or a doubling.  The plain sequence (lf2, one
    // turning() pass per criterion and level, lf1) is a chain of up to 3 * depth + 4 dependent passes over the row.  Here
    //   * ONE pass takes the second half of the leapfrog, the kinetic energy, the level-0 criterion AND the criteria of the first
    //     merge level >= 1 the leaf closes (which sub-trees a leaf closes, and in which P-slots their summaries lie, follows
    //     from the leaf number alone), and — speculatively — the FIRST half of the next leapfrog: the next leaf of a doubling
    //     always continues from this one; when the tree ends instead, what was written is dead storage;
    //   * deeper levels and the top-level merge take one more pass per two levels, against the new leaf kept in registers;
    //   * the scalar cascade (cont_tree: weights, multinomial merges) then only consumes bits.
    // Each criterion keeps its own pair of accumulators, summed per (lane, component) over the wave's chunks in increasing
    // order and reduced in the contract's order — the bits of the plain sequence, which remains the fall-back (rows of more
    // than CBK chunks per wave; evaluations that failed).
    // ======================================================================================
#ifndef NPHIP_CB_CHUNKS
#define NPHIP_CB_CHUNKS 2
#endif
    static constexpr int CBK = NPHIP_CB_CHUNKS;   // chunks per wave the fused leaf holds in registers (host.hip: choose_waves_callback)
    static constexpr int kParkMax = 3 + 6 * (kMaxDepthCap - 1) + 6;   // K, level 0, six values per level >= 1 and for the top-level merge
    struct CbCrit {
        uint32_t bits;        // bit k: the criteria of merge level k say "turning"  (level 0 = the pair (source, new leaf))
        bool turn_top;        // the criteria of the top-level merge (only meaningful for the leaf that completes the doubling)
        int64_t spec_q, spec_p;   // >= 0: the first half of the NEXT leapfrog has been taken into these buffers
    };
    __device__ __forceinline__ bool cb_fast_ok() const { return !INK && NV == 0 && nch <= (int64_t)CBK * W; }
    // P-slots of A.first / A.last of the sub-tree that waits at merge level k (>= 1) when leaf j of a doubling of depth d arrives
    __device__ __forceinline__ void level_slots(int64_t j, int64_t d, int k, int64_t& sA, int64_t& sB) const {
        const int64_t a = j - (2ll << k) + 1, al = j - (1ll << k);
        sA = (a == 1) ? slot_first((int)d) : slot_first(__builtin_ctzll((unsigned long long)(a - 1)));
        sB = slot_last(__builtin_ctzll((unsigned long long)al), A.cap);
    }
    struct CbRegs { double2 tp[CBK], tr[CBK]; };   // T.last = the new leaf: (p, rho)
    // the fused pass; L = merge levels >= 1 whose criteria ride along (0, 1).  out = K, level-0 pair, 6 sums per level
    template <int L>
    __device__ __forceinline__ void cb_pass1(int64_t newq, int64_t newp, int64_t srcp, int64_t nextq, int64_t nextp, double h, double eps, bool copy_rho,
                                             const int64_t (&sA)[2], const int64_t (&sB)[2], CbRegs& T_, double (&out)[3 + 6 * L]) {
        constexpr int LX = L > 0 ? L : 1;
        double2 acc[3 + 6 * L];
#pragma unroll
        for (int n = 0; n < 3 + 6 * L; ++n) { acc[n].x = 0.0; acc[n].y = 0.0; }
        const double* ge = A.geval + (size_t)chain * D;
        double *g = G(newq), *pn = P(newp), *rn = R(newp);
        const double *rp = R(srcp), *ps = P(srcp), *qn = Q(newq);
        double2 gv[CBK], ph[CBK], s2[CBK], r2[CBK], p2[CBK], q2[CBK], ap[CBK][LX], ar[CBK][LX], bp[CBK][LX], br[CBK][LX];
        // every read of the pass first (both chunks): one round trip
#pragma unroll
        for (int k = 0; k < CBK; ++k) {
            int64_t cc = wave + (int64_t)k * W;
            cc = cc < nch ? cc : nch - 1;   // (past the end: the last chunk again, never used — unconditional reads stay in registers)
            const int64_t i = cc * NPHIP_CHUNK + 2 * lane;
            const uint32_t ob = (uint32_t)i * 8u;
            gv[k] = ld2_dense(ge, i, D); ph[k] = ld2b(pn, ob); s2[k] = ld2b(sig2, ob); r2[k] = ld2b(rp, ob); p2[k] = ld2b(ps, ob);
            if (nextq >= 0) q2[k] = ld2b(qn, ob);
#pragma unroll
            for (int l = 0; l < L; ++l) { ap[k][l] = ld2b(P(sA[l]), ob); ar[k][l] = ld2b(R(sA[l]), ob); bp[k][l] = ld2b(P(sB[l]), ob); br[k][l] = ld2b(R(sB[l]), ob); }
        }
#pragma unroll
        for (int k = 0; k < CBK; ++k) {
            const int64_t cc = wave + (int64_t)k * W;
            if (cc < nch) {
                const int64_t i = cc * NPHIP_CHUNK + 2 * lane;
                const uint32_t ob = (uint32_t)i * 8u;
                double2 pv, rr;
                pv.x = fma(h, gv[k].x, ph[k].x);
                pv.y = fma(h, gv[k].y, ph[k].y);
                const double vx = s2[k].x * pv.x, vy = s2[k].y * pv.y;
                acc[0].x = fma(pv.x, vx, acc[0].x);
                acc[0].y = fma(pv.y, vy, acc[0].y);
                rr.x = copy_rho ? pv.x : r2[k].x + pv.x;
                rr.y = copy_rho ? pv.y : r2[k].y + pv.y;
                // level 0: the pair (source, new leaf) — turning() on the two P-slots, from the registers
                const double tx0 = (rr.x - r2[k].x) + p2[k].x, ty0 = (rr.y - r2[k].y) + p2[k].y;
                acc[1].x = fma(tx0, vx, acc[1].x);
                acc[1].y = fma(ty0, vy, acc[1].y);
                acc[2].x = fma(tx0, s2[k].x * p2[k].x, acc[2].x);
                acc[2].y = fma(ty0, s2[k].y * p2[k].y, acc[2].y);
#pragma unroll
                for (int l = 0; l < L; ++l) {
                    const double2 fpv = l == 0 ? p2[k] : ap[k][l > 0 ? l - 1 : 0], frv = l == 0 ? r2[k] : ar[k][l > 0 ? l - 1 : 0];   // T.first of this level
                    span_acc(ap[k][l].x, ar[k][l].x, pv.x, rr.x, s2[k].x, acc[3 + 6 * l].x, acc[4 + 6 * l].x);      // (A.first, T.last)
                    span_acc(ap[k][l].y, ar[k][l].y, pv.y, rr.y, s2[k].y, acc[3 + 6 * l].y, acc[4 + 6 * l].y);
                    span_acc(bp[k][l].x, br[k][l].x, pv.x, rr.x, s2[k].x, acc[5 + 6 * l].x, acc[6 + 6 * l].x);      // (A.last, T.last)
                    span_acc(bp[k][l].y, br[k][l].y, pv.y, rr.y, s2[k].y, acc[5 + 6 * l].y, acc[6 + 6 * l].y);
                    span_acc(ap[k][l].x, ar[k][l].x, fpv.x, frv.x, s2[k].x, acc[7 + 6 * l].x, acc[8 + 6 * l].x);    // (A.first, T.first)
                    span_acc(ap[k][l].y, ar[k][l].y, fpv.y, frv.y, s2[k].y, acc[7 + 6 * l].y, acc[8 + 6 * l].y);
                }
                st2b(g, ob, gv[k]);
                st2b(pn, ob, pv);
                st2b(rn, ob, rr);
                if (nextq >= 0) {   // first half of the next leapfrog (lf1), continuing from this leaf
                    double2 ph2, qq;
                    ph2.x = fma(h, gv[k].x, pv.x);
                    ph2.y = fma(h, gv[k].y, pv.y);
                    qq.x = fma(eps, s2[k].x * ph2.x, q2[k].x);
                    qq.y = fma(eps, s2[k].y * ph2.y, q2[k].y);
                    st2b(Q(nextq), ob, qq);
                    st2b(P(nextp), ob, ph2);
                    qe_st(i, qq);
                }
                T_.tp[k] = pv; T_.tr[k] = rr;
            }
        }
#pragma unroll
        for (int n = 0; n < 3 + 6 * L; ++n) out[n] = acc[n].x + acc[n].y;
        wave_sumN(out);   // (the wave's totals: the waves of the chain are added once, after the last pass — leaf_cb)
    }
    // criteria of merge level k (>= 1) against the new leaf in registers: the wave's totals.  (A.first, T.last) || (A.last, T.last) ||
    // (A.first, T.first); T.first = the first leaf of T: the source of this leapfrog at level 1, A.first of the level below otherwise
    __device__ __forceinline__ void cb_level(int64_t j, int64_t d, int k, int64_t srcp, const CbRegs& T_, double (&v)[6]) {
        int64_t sA, sB, sF = srcp, unused;
        level_slots(j, d, k, sA, sB);
        if (k > 1) level_slots(j, d, k - 1, sF, unused);
        double2 acc[6], ap[CBK], ar[CBK], bp[CBK], br[CBK], fp[CBK], fr[CBK], sg[CBK];
#pragma unroll
        for (int n = 0; n < 6; ++n) { acc[n].x = 0.0; acc[n].y = 0.0; }
#pragma unroll
        for (int q_ = 0; q_ < CBK; ++q_) {
            int64_t cc = wave + (int64_t)q_ * W;
            cc = cc < nch ? cc : nch - 1;
            const int64_t i = cc * NPHIP_CHUNK + 2 * lane;
            const uint32_t ob = (uint32_t)i * 8u;
            ap[q_] = ld2b(P(sA), ob); ar[q_] = ld2b(R(sA), ob); bp[q_] = ld2b(P(sB), ob); br[q_] = ld2b(R(sB), ob);
            fp[q_] = ld2b(P(sF), ob); fr[q_] = ld2b(R(sF), ob); sg[q_] = ld2b(sig2, ob);
        }
#pragma unroll
        for (int q_ = 0; q_ < CBK; ++q_) {
            if (wave + (int64_t)q_ * W < nch) {
                span_acc(ap[q_].x, ar[q_].x, T_.tp[q_].x, T_.tr[q_].x, sg[q_].x, acc[0].x, acc[1].x);
                span_acc(ap[q_].y, ar[q_].y, T_.tp[q_].y, T_.tr[q_].y, sg[q_].y, acc[0].y, acc[1].y);
                span_acc(bp[q_].x, br[q_].x, T_.tp[q_].x, T_.tr[q_].x, sg[q_].x, acc[2].x, acc[3].x);
                span_acc(bp[q_].y, br[q_].y, T_.tp[q_].y, T_.tr[q_].y, sg[q_].y, acc[2].y, acc[3].y);
                span_acc(ap[q_].x, ar[q_].x, fp[q_].x, fr[q_].x, sg[q_].x, acc[4].x, acc[5].x);
                span_acc(ap[q_].y, ar[q_].y, fp[q_].y, fr[q_].y, sg[q_].y, acc[4].y, acc[5].y);
            }
        }
#pragma unroll
        for (int n = 0; n < 6; ++n) v[n] = acc[n].x + acc[n].y;
        wave_sumN(v);
    }
    // criteria of the top-level merge (the leaf that completes the doubling), general index modes: (far, T.last) || (near, T.last) ||
    // (far, T.first); depth 0: (far, T.last) alone — far and near end are the origin
    __device__ __forceinline__ void cb_top(int64_t d, int64_t far_slot, int64_t far_idx, int64_t near_slot, int64_t near_idx, int64_t dir, int64_t idx_new,
                                           const CbRegs& T_, double (&v)[6]) {
        const Pair p1 = pair_of(far_idx, idx_new), p2 = pair_of(near_idx, idx_new), p3 = pair_of(far_idx, near_idx + dir);
        const int64_t sF = slot_first((int)d);   // T.first = leaf 1 of the doubling
        double2 acc[6], fa[CBK], fb[CBK], na[CBK], nb[CBK], tfp[CBK], tfr[CBK], sg[CBK];
#pragma unroll
        for (int n = 0; n < 6; ++n) { acc[n].x = 0.0; acc[n].y = 0.0; }
#pragma unroll
        for (int k = 0; k < CBK; ++k) {
            int64_t cc = wave + (int64_t)k * W;
            cc = cc < nch ? cc : nch - 1;
            const int64_t i = cc * NPHIP_CHUNK + 2 * lane;
            const uint32_t ob = (uint32_t)i * 8u;
            fa[k] = ld2b(P(far_slot), ob); fb[k] = ld2b(R(far_slot), ob); sg[k] = ld2b(sig2, ob);
            if (d > 0) { na[k] = ld2b(P(near_slot), ob); nb[k] = ld2b(R(near_slot), ob); tfp[k] = ld2b(P(sF), ob); tfr[k] = ld2b(R(sF), ob); }
        }
#pragma unroll
        for (int k = 0; k < CBK; ++k) {
            if (wave + (int64_t)k * W < nch) {
                pair_acc(p1, fa[k].x, fb[k].x, T_.tp[k].x, T_.tr[k].x, sg[k].x, acc[0].x, acc[1].x);
                pair_acc(p1, fa[k].y, fb[k].y, T_.tp[k].y, T_.tr[k].y, sg[k].y, acc[0].y, acc[1].y);
                if (d > 0) {
                    pair_acc(p2, na[k].x, nb[k].x, T_.tp[k].x, T_.tr[k].x, sg[k].x, acc[2].x, acc[3].x);
                    pair_acc(p2, na[k].y, nb[k].y, T_.tp[k].y, T_.tr[k].y, sg[k].y, acc[2].y, acc[3].y);
                    pair_acc(p3, fa[k].x, fb[k].x, tfp[k].x, tfr[k].x, sg[k].x, acc[4].x, acc[5].x);
                    pair_acc(p3, fa[k].y, fb[k].y, tfp[k].y, tfr[k].y, sg[k].y, acc[4].y, acc[5].y);
                }
            }
        }
#pragma unroll
        for (int n = 0; n < 6; ++n) v[n] = acc[n].x + acc[n].y;
        wave_sumN(v);
    }
    // Deferred chain-wide sums of the fused leaf: every group of values is reduced within the wave as soon as its pass is through
    // and parked in LDS (wave totals, [value][wave]); after the LAST pass one barrier, and every wave adds the wave totals of every
    // group in wave order — the bits of one rsum() per group, for one barrier instead of one per pass.  (One wave per chain: the
    // wave totals are the sums.)
    template <int N>
    __device__ __forceinline__ void park(const double (&v)[N], int at) {
        if (W > 1 && lane == 0) {
#pragma unroll
            for (int n = 0; n < N; ++n) parked[(at + n) * W + wave] = v[n];
        }
    }
    __device__ __forceinline__ double parked_sum(int at) const {
        double t = parked[at * W];
        for (int w = 1; w < W; ++w) t = t + parked[at * W + w];
        return t;
    }
    __device__ __forceinline__ bool parked_any_negative(int at, int n) const {
        bool t = false;
        for (int i = 0; i < n; ++i) t = t || (parked_sum(at + i) < 0.0);
        return t;
    }
    template <int N>
    static __device__ __forceinline__ bool any_negative(const double (&v)[N], int from) {
        bool t = false;
#pragma unroll
        for (int n = 0; n < N; ++n) t = t || (n >= from && v[n] < 0.0);
        return t;
    }
    // The evaluation of the pending tree leapfrog has arrived (finite, code 0): everything cont_tree needs, in as few dependent
    // passes as the leaf allows.  Returns the kinetic energy.
    __device__ __forceinline__ double leaf_cb(CbCrit& cr) {
        // (the control words are the same in every lane: as scalars, so that slots and addresses are scalar too)
        const int64_t j = rfl(c->nleaf) + 1, d = rfl(c->depth), dir = rfl(c->dir);
        const int db = dir > 0 ? 1 : 0;
        const int64_t idx_new = rfl(c->idx_cur) + dir;
        const int64_t newq = rfl(c->lf_newq), newp = rfl(c->lf_newp), srcp = rfl(c->lf_srcp);
        const double eps = (double)dir * rfl_f64(c->step_size), h = 0.5 * eps;
        const bool complete = j == (1ll << d);
        const bool check = A.s.check_turning && (d + 1 > A.s.mindepth);
        // merge levels this leaf closes: k = 0 .. nm - 1  (bits 0 .. k of j - 1 set, k < d)
        int nm = __builtin_ctzll(~(unsigned long long)(j - 1));
        nm = nm < (int)d ? nm : (int)d;
        const int nlev = check && nm > 1 ? nm - 1 : 0;   // levels >= 1 with criteria
        // the next leaf (unless this one completes the doubling): its P-slot by its number, its Q-pool buffer = any that neither
        // the tree nor this leaf holds (the cursor's own buffer is free once the leapfrog that started from it is done — but the
        // divergence record reads the state a failed leapfrog started from: no speculation then)
        cr.spec_q = -1; cr.spec_p = -1;
        if (!complete && A.tr_div[0] == nullptr) {
            const int64_t jn = j + 1;
            if (jn == (1ll << d)) cr.spec_p = slot_end(db, (int)(rfl(c->endpar[db]) ^ 1));
            else if (jn & 1) cr.spec_p = slot_first(__builtin_ctzll((unsigned long long)(jn - 1)));
            else cr.spec_p = slot_last(__builtin_ctzll((unsigned long long)jn), A.cap);
            uint32_t used = (1u << rfl(c->cand_q)) | (1u << rfl(c->endq[0])) | (1u << rfl(c->endq[1])) | (1u << newq);
            for (uint64_t m = (uint64_t)(j - 1) & ((1ull << kMaxDepthCap) - 1); m != 0; m &= m - 1) used |= 1u << rfl(c->sub_q[__builtin_ctzll(m)]);
            cr.spec_q = (int64_t)__builtin_ctz(~used);
        }
        int64_t sA[2] = {0, 0}, sB[2] = {0, 0};
        CbRegs T_;
        double K = 0.0;
        cr.bits = 0; cr.turn_top = false;
        const bool copy_rho = idx_new == -1;
        // parked values: [0] K, [1, 2] level 0, then six per level >= 1, then the six of the top-level merge
        {
            double v[3];
            cb_pass1<0>(newq, newp, srcp, cr.spec_q, cr.spec_p, h, eps, copy_rho, sA, sB, T_, v);
            if (W == 1) { K = 0.5 * v[0]; cr.bits = ((v[1] < 0.0) || (v[2] < 0.0)) ? 1u : 0u; }
            else park(v, 0);
        }
        // levels >= 1, one pass each (operands mostly in L2; no stores in between, so their reads overlap; ONE barrier at the end).
        // Two levels per pass or level 1 inside the fused pass need 150+ registers per lane: two waves per SIMD instead of four,
        // i.e. half of a 1024-chain batch waiting for the other half (measured: 44 against 37 us per step)
        for (int k0 = 1; k0 <= nlev; ++k0) {
            double u[6];
            cb_level(j, d, k0, srcp, T_, u);
            if (W == 1) cr.bits |= any_negative(u, 0) ? (1u << k0) : 0u;
            else park(u, 3 + 6 * (k0 - 1));
        }
        const bool top = complete && check;
        const int at_top = 3 + 6 * nlev;
        if (top) {
            const int64_t far_idx = rfl(dir > 0 ? c->idx_left : c->idx_right), near_idx = rfl(dir > 0 ? c->idx_right : c->idx_left);
            double u[6];
            cb_top(d, rfl(c->endp[1 - db]), far_idx, rfl(c->endp[db]), near_idx, dir, idx_new, T_, u);
            if (W == 1) cr.turn_top = any_negative(u, 0);
            else park(u, at_top);
        }
        if (W > 1) {
            __syncthreads();
            K = 0.5 * parked_sum(0);
            cr.bits = ((parked_sum(1) < 0.0) || (parked_sum(2) < 0.0)) ? 1u : 0u;
            for (int k = 1; k <= nlev; ++k) cr.bits |= parked_any_negative(3 + 6 * (k - 1), 6) ? (1u << k) : 0u;
            if (top) cr.turn_top = parked_any_negative(at_top, 6);
        }
        if (!check) cr.bits = 0;
        return K;
    }

    // ---- register-resident leapfrog (NV > 0): one fused pass, no loads when continuing from the cursor ----
    // element pair of this lane in the k-th chunk this wave owns (chunk c belongs to wave c mod W)
    __device__ __forceinline__ int64_t ridx(int k) const { return ((int64_t)k * W + wave) * NPHIP_CHUNK + 2 * lane; }


    // ---- model parameters / chunk-edge neighbours of the register kernels --------------------------------------
    // W == 1: parameters from the LDS copy, edges by v_readlane of the neighbouring chunk's registers.
    // W  > 1: parameters from global memory (L2-resident, shared by all chains), edges through a small LDS buffer:
    //         every wave publishes the first and last z of its chunks, one workgroup barrier, neighbours read back.
    __device__ __forceinline__ double2 par_mu(int64_t i) const {
        if (W == 1) return *(const NPHIP_LDS double2*)(par + i);
        return ld2(A.m_mu, i);
    }
    __device__ __forceinline__ void par_ab(int64_t i, double2& a, double2& b01, double& b2) const {
        if (W == 1) {
            a = *(const NPHIP_LDS double2*)(par + ld + i);
            b01 = *(const NPHIP_LDS double2*)(par + 2 * ld + i);   // b_{i-1}, b_i
            b2 = par[2 * ld + i + 2];                               // b_{i+1}
        } else {
            a = ld2(A.m_a, i);
            b01 = ld2(A.m_bsh, i);
            b2 = ld1(A.m_bsh, i + 2);
        }
    }
    __device__ __forceinline__ void publish_edges(const double2 (&z)[NVX]) {
        if (W == 1) return;
#pragma unroll
        for (int k = 0; k < NVX; ++k) {
            const int64_t cch = (int64_t)k * W + wave;
            if (lane == 0) edge[2 * cch] = z[k].x;
            if (lane == 63) edge[2 * cch + 1] = z[k].y;
        }
        __syncthreads();
    }
    __device__ __forceinline__ void edge_pair(const double2 (&z)[NVX], int k, double& zl, double& zr) const {
        zl = 0.0; zr = 0.0;
        if (W == 1) {
            if (k > 0) zl = readlane_f64(z[k > 0 ? k - 1 : 0].y, 63);
            if (k + 1 < NVX) zr = readlane_f64(z[k + 1 < NVX ? k + 1 : 0].x, 0);
        } else {
            const int64_t cch = (int64_t)k * W + wave;
            if (cch > 0) zl = edge[2 * (cch - 1) + 1];
            if (cch + 1 < nch) zr = edge[2 * (cch + 1)];
        }
    }

    // ======================================================================================
    // Register-resident leaf (NV > 0): leapfrog + merge cascade + stores of ONE tree leaf.
    //
    // Ordering is chosen for gfx9's single vmcnt counter (loads and stores share it):
    //   1. prefetch the level-1 U-turn operands (only when this leaf will reach level 1),
    //   2. leapfrog math — model parameters come from LDS, neighbours from DPP wave shifts; the level-0
    //      criterion is accumulated in the same pass from the before/after registers,
    //   3. merge cascade (further operands on demand; T.first of level k is A.first of level k-1 and is reused),
    //   4. stores of the new state LAST, and only of what can be read back later:
    //        (q, grad): when the leaf is referenced as a draw candidate or becomes a trajectory end,
    //        (p, rho) : unless leaf % 4 == 3 (such a leaf is only ever "T.first of level 1" = the source
    //                   registers of the next leapfrog).
    //      Elided parts are marked dirty and flushed before any out-of-line path / at kernel exit.
    // ======================================================================================
    struct Pair { int mode; bool first_is_start; };
    __device__ __forceinline__ Pair pair_of(int64_t i1, int64_t i2) const {
        Pair pr;
        pr.first_is_start = i1 < i2;
        const int64_t a = pr.first_is_start ? i1 : i2, b = pr.first_is_start ? i2 : i1;
        pr.mode = (a >= 0 && b >= 0) ? 0 : ((b >= 0 && a < 0) ? 1 : 2);
        return pr;
    }
    // accumulate span . v_end (acc_e) and span . v_start (acc_s) for the pair (1, 2)   
    __device__ __forceinline__ void pair_acc(const Pair pr, double p1, double r1, double p2, double r2, double s2v, double& acc_e, double& acc_s) const {
        const double ps = pr.first_is_start ? p1 : p2, rs = pr.first_is_start ? r1 : r2;
        const double pe = pr.first_is_start ? p2 : p1, re = pr.first_is_start ? r2 : r1;
        double t;
        if (pr.mode == 0) t = (re - rs) + ps;
        else if (pr.mode == 1) t = re + rs;
        else t = (rs - re) + pe;
        acc_e = fma(t, s2v * pe, acc_e);
        acc_s = fma(t, s2v * ps, acc_s);
    }
    __device__ __forceinline__ void pair_acc_v(const Pair pr, double p1, double r1, double v1, double p2, double r2, double v2, double& acc_e, double& acc_s) const {
        const double ps = pr.first_is_start ? p1 : p2, rs = pr.first_is_start ? r1 : r2, vs = pr.first_is_start ? v1 : v2;
        const double pe = pr.first_is_start ? p2 : p1, re = pr.first_is_start ? r2 : r1, ve = pr.first_is_start ? v2 : v1;
        double t;
        if (pr.mode == 0) t = (re - rs) + ps;
        else if (pr.mode == 1) t = re + rs;
        else t = (rs - re) + pe;
        acc_e = fma(t, ve, acc_e);
        acc_s = fma(t, vs, acc_s);
    }
    __device__ __forceinline__ void load_slot_v(int64_t slot, double2 (&v)[NVX]) const {
        const double* gv = VEL(slot);
#pragma unroll
        for (int k = 0; k < NVX; ++k) v[k] = ld2(gv, ridx(k));
    }
    __device__ __forceinline__ void load_slot(int64_t slot, double2 (&p)[NVX], double2 (&r)[NVX]) const {
        const double *gp = P(slot), *gr = R(slot);
        constexpr int nk = NVX;  // the register kernels are instantiated per exact chunk count
#pragma unroll
        for (int k = 0; k < NVX; ++k) if (k < nk) { p[k] = ld2(gp, ridx(k)); r[k] = ld2(gr, ridx(k)); }
    }
    __device__ __forceinline__ int64_t first_slot_of(int64_t leaf, int64_t d) const {
        return (leaf == 1) ? slot_first((int)d) : slot_first(__builtin_ctzll((unsigned long long)(leaf - 1)));
    }
    // (A, TL) || (A, TF): four dots in one pass
    __device__ __forceinline__ bool check_a(const RegsT& X, const double2 (&ap)[NVX], const double2 (&ar)[NVX], const double2 (&fp)[NVX],
                                            const double2 (&fr)[NVX], int64_t iA, int64_t iTF, int64_t iTL) {
        constexpr int nk = NVX;  // the register kernels are instantiated per exact chunk count
        const Pair p1 = pair_of(iA, iTL), p3 = pair_of(iA, iTF);
        double2 acc[4];
#pragma unroll
        for (int n = 0; n < 4; ++n) { acc[n].x = 0.0; acc[n].y = 0.0; }
#pragma unroll
        for (int k = 0; k < NVX; ++k) if (k < nk) {
            pair_acc(p1, ap[k].x, ar[k].x, X.p[k].x, X.r[k].x, X.s[k].x, acc[0].x, acc[1].x);
            pair_acc(p1, ap[k].y, ar[k].y, X.p[k].y, X.r[k].y, X.s[k].y, acc[0].y, acc[1].y);
            pair_acc(p3, ap[k].x, ar[k].x, fp[k].x, fr[k].x, X.s[k].x, acc[2].x, acc[3].x);
            pair_acc(p3, ap[k].y, ar[k].y, fp[k].y, fr[k].y, X.s[k].y, acc[2].y, acc[3].y);
        }
        double v[4];
#pragma unroll
        for (int n = 0; n < 4; ++n) v[n] = acc[n].x + acc[n].y;
        rsum(v);
        return (v[0] < 0.0) || (v[1] < 0.0) || (v[2] < 0.0) || (v[3] < 0.0);
    }
    __device__ __forceinline__ bool check1(const RegsT& X, const double2 (&ap)[NVX], const double2 (&ar)[NVX], int64_t iA, int64_t iTL) {
        constexpr int nk = NVX;  // the register kernels are instantiated per exact chunk count
        const Pair p1 = pair_of(iA, iTL);
        double2 e = {0.0, 0.0}, st = {0.0, 0.0};
#pragma unroll
        for (int k = 0; k < NVX; ++k) if (k < nk) {
            pair_acc(p1, ap[k].x, ar[k].x, X.p[k].x, X.r[k].x, X.s[k].x, e.x, st.x);
            pair_acc(p1, ap[k].y, ar[k].y, X.p[k].y, X.r[k].y, X.s[k].y, e.y, st.y);
        }
        double v[2] = {e.x + e.y, st.x + st.y};
        rsum(v);
        return (v[0] < 0.0) || (v[1] < 0.0);
    }
    // pass A: (A.first, TL) || (A.first, TF) with TF in registers
    __device__ __forceinline__ bool sub_a(const RegsT& X, const double2 (&ap)[NVX], const double2 (&ar)[NVX], const double2 (&fp)[NVX], const double2 (&fr)[NVX]) {
        constexpr int nk = NVX;  // the register kernels are instantiated per exact chunk count
        double2 acc[4];
#pragma unroll
        for (int n = 0; n < 4; ++n) { acc[n].x = 0.0; acc[n].y = 0.0; }
#pragma unroll
        for (int k = 0; k < NVX; ++k) if (k < nk) {
            span_acc(ap[k].x, ar[k].x, X.p[k].x, X.r[k].x, X.s[k].x, acc[0].x, acc[1].x);
            span_acc(ap[k].y, ar[k].y, X.p[k].y, X.r[k].y, X.s[k].y, acc[0].y, acc[1].y);
            span_acc(ap[k].x, ar[k].x, fp[k].x, fr[k].x, X.s[k].x, acc[2].x, acc[3].x);
            span_acc(ap[k].y, ar[k].y, fp[k].y, fr[k].y, X.s[k].y, acc[2].y, acc[3].y);
        }
        double v[4];
#pragma unroll
        for (int n = 0; n < 4; ++n) v[n] = acc[n].x + acc[n].y;
        rsum(v);
        return (v[0] < 0.0) || (v[1] < 0.0) || (v[2] < 0.0) || (v[3] < 0.0);
    }
    // pass B: (A.last, TL)
    __device__ __forceinline__ bool sub_b(const RegsT& X, const double2 (&ap)[NVX], const double2 (&ar)[NVX]) {
        constexpr int nk = NVX;  // the register kernels are instantiated per exact chunk count
        double2 e = {0.0, 0.0}, st = {0.0, 0.0};
#pragma unroll
        for (int k = 0; k < NVX; ++k) if (k < nk) {
            span_acc(ap[k].x, ar[k].x, X.p[k].x, X.r[k].x, X.s[k].x, e.x, st.x);
            span_acc(ap[k].y, ar[k].y, X.p[k].y, X.r[k].y, X.s[k].y, e.y, st.y);
        }
        double v[2] = {e.x + e.y, st.x + st.y};
        rsum(v);
        return (v[0] < 0.0) || (v[1] < 0.0);
    }
    // ---- the same four under the low-rank metric (LR): every operand comes with its velocity
    __device__ __forceinline__ bool check_a_v(const RegsT& X, const double2 (&ap)[NVX], const double2 (&ar)[NVX], const double2 (&av)[NVX], const double2 (&fp)[NVX],
                                              const double2 (&fr)[NVX], const double2 (&fv)[NVX], int64_t iA, int64_t iTF, int64_t iTL) {
        const Pair p1 = pair_of(iA, iTL), p3 = pair_of(iA, iTF);
        double2 acc[4];
#pragma unroll
        for (int n = 0; n < 4; ++n) { acc[n].x = 0.0; acc[n].y = 0.0; }
#pragma unroll
        for (int k = 0; k < NVX; ++k) {
            pair_acc_v(p1, ap[k].x, ar[k].x, av[k].x, X.p[k].x, X.r[k].x, X.v[k].x, acc[0].x, acc[1].x);
            pair_acc_v(p1, ap[k].y, ar[k].y, av[k].y, X.p[k].y, X.r[k].y, X.v[k].y, acc[0].y, acc[1].y);
            pair_acc_v(p3, ap[k].x, ar[k].x, av[k].x, fp[k].x, fr[k].x, fv[k].x, acc[2].x, acc[3].x);
            pair_acc_v(p3, ap[k].y, ar[k].y, av[k].y, fp[k].y, fr[k].y, fv[k].y, acc[2].y, acc[3].y);
        }
        double v[4];
#pragma unroll
        for (int n = 0; n < 4; ++n) v[n] = acc[n].x + acc[n].y;
        rsum(v);
        return (v[0] < 0.0) || (v[1] < 0.0) || (v[2] < 0.0) || (v[3] < 0.0);
    }
    __device__ __forceinline__ bool check1_v(const RegsT& X, const double2 (&ap)[NVX], const double2 (&ar)[NVX], const double2 (&av)[NVX], int64_t iA, int64_t iTL) {
        const Pair p1 = pair_of(iA, iTL);
        double2 e = {0.0, 0.0}, st = {0.0, 0.0};
#pragma unroll
        for (int k = 0; k < NVX; ++k) {
            pair_acc_v(p1, ap[k].x, ar[k].x, av[k].x, X.p[k].x, X.r[k].x, X.v[k].x, e.x, st.x);
            pair_acc_v(p1, ap[k].y, ar[k].y, av[k].y, X.p[k].y, X.r[k].y, X.v[k].y, e.y, st.y);
        }
        double v[2] = {e.x + e.y, st.x + st.y};
        rsum(v);
        return (v[0] < 0.0) || (v[1] < 0.0);
    }
    __device__ __forceinline__ bool sub_a_v(const RegsT& X, const double2 (&ap)[NVX], const double2 (&ar)[NVX], const double2 (&av)[NVX], const double2 (&fp)[NVX],
                                            const double2 (&fr)[NVX], const double2 (&fv)[NVX]) {
        double2 acc[4];
#pragma unroll
        for (int n = 0; n < 4; ++n) { acc[n].x = 0.0; acc[n].y = 0.0; }
#pragma unroll
        for (int k = 0; k < NVX; ++k) {
            span_acc_v(ap[k].x, ar[k].x, av[k].x, X.r[k].x, X.v[k].x, acc[0].x, acc[1].x);
            span_acc_v(ap[k].y, ar[k].y, av[k].y, X.r[k].y, X.v[k].y, acc[0].y, acc[1].y);
            span_acc_v(ap[k].x, ar[k].x, av[k].x, fr[k].x, fv[k].x, acc[2].x, acc[3].x);
            span_acc_v(ap[k].y, ar[k].y, av[k].y, fr[k].y, fv[k].y, acc[2].y, acc[3].y);
        }
        double v[4];
#pragma unroll
        for (int n = 0; n < 4; ++n) v[n] = acc[n].x + acc[n].y;
        rsum(v);
        return (v[0] < 0.0) || (v[1] < 0.0) || (v[2] < 0.0) || (v[3] < 0.0);
    }
    __device__ __forceinline__ bool sub_b_v(const RegsT& X, const double2 (&ap)[NVX], const double2 (&ar)[NVX], const double2 (&av)[NVX]) {
        double2 e = {0.0, 0.0}, st = {0.0, 0.0};
#pragma unroll
        for (int k = 0; k < NVX; ++k) {
            span_acc_v(ap[k].x, ar[k].x, av[k].x, X.r[k].x, X.v[k].x, e.x, st.x);
            span_acc_v(ap[k].y, ar[k].y, av[k].y, X.r[k].y, X.v[k].y, e.y, st.y);
        }
        double v[2] = {e.x + e.y, st.x + st.y};
        rsum(v);
        return (v[0] < 0.0) || (v[1] < 0.0);
    }
    // ---- LDS ring of recent (p, rho) summaries
    __device__ __forceinline__ NPHIP_LDS double2* ring_ptr(int slot, int vec) const {
        return (NPHIP_LDS double2*)(ring + (size_t)(slot * 2 + vec) * NVX * 128) + lane;
    }
    __device__ __forceinline__ void ring_write(int slot, const double2 (&p)[NVX], const double2 (&r)[NVX]) {
        constexpr int nk = NVX;  // the register kernels are instantiated per exact chunk count
        NPHIP_LDS double2 *lp = ring_ptr(slot, 0), *lr = ring_ptr(slot, 1);
#pragma unroll
        for (int k = 0; k < NVX; ++k) if (k < nk) { lp[k * 64] = p[k]; lr[k * 64] = r[k]; }
    }
    __device__ __forceinline__ void ring_read(int slot, double2 (&p)[NVX], double2 (&r)[NVX]) const {
        constexpr int nk = NVX;  // the register kernels are instantiated per exact chunk count
        const NPHIP_LDS double2 *lp = ring_ptr(slot, 0), *lr = ring_ptr(slot, 1);
#pragma unroll
        for (int k = 0; k < NVX; ++k) if (k < nk) { p[k] = lp[k * 64]; r[k] = lr[k * 64]; }
    }
    // fused-model gradient of the register position (used when a position is reloaded: only q is kept in HBM)
    __device__ __forceinline__ void regs_grad(RegsT& X) {
        double2 z[NVX];
#pragma unroll
        for (int k = 0; k < NVX; ++k) {
            const double2 mu = par_mu(ridx(k));
            z[k].x = X.q[k].x - mu.x;
            z[k].y = X.q[k].y - mu.y;
        }
        publish_edges(z);
#pragma unroll
        for (int k = 0; k < NVX; ++k) {
            double2 a, b01;
            double b2, edge_zl, edge_zr;
            par_ab(ridx(k), a, b01, b2);
            edge_pair(z, k, edge_zl, edge_zr);
            const double zl = wave_shr1(z[k].y, edge_zl), zr = wave_shl1(z[k].x, edge_zr);
            double tx = a.x * z[k].x;
            tx = fma(b01.x, zl, tx);
            tx = fma(b01.y, z[k].y, tx);
            double ty = a.y * z[k].y;
            ty = fma(b01.y, z[k].x, ty);
            ty = fma(b2, zr, ty);
            X.g[k].x = -tx;
            X.g[k].y = -ty;
        }
        if (W > 1) __syncthreads();  // the edge buffer is free again
    }
    // HBM copies: q only (the gradient is recomputed on reload), (p, rho) into the leaf's P-slot
    __device__ __forceinline__ void store_state(RegsT& X, bool q_, bool pr) {
        constexpr int nk = NVX;  // the register kernels are instantiated per exact chunk count
        if (q_) {
            double* qn = Q(X.reg_q);
#pragma unroll
            for (int k = 0; k < NVX; ++k) if (k < nk) st2(qn, ridx(k), X.q[k]);
            if (REMOTE) {   // a callback model's gradient cannot be rebuilt on reload: it is stored with the position
                double* gn = G(X.reg_q);
#pragma unroll
                for (int k = 0; k < NVX; ++k) if (k < nk) st2(gn, ridx(k), X.g[k]);
            }
            X.dirty_qg = false;
        }
        if (pr) {
            double *pn = P(X.reg_p), *rn = R(X.reg_p);
#pragma unroll
            for (int k = 0; k < NVX; ++k) if (k < nk) { st2(pn, ridx(k), X.p[k]); st2(rn, ridx(k), X.r[k]); }
            if (LR) {
                double* vn = VEL(X.reg_p);
#pragma unroll
                for (int k = 0; k < NVX; ++k) st2(vn, ridx(k), X.v[k]);
            }
            X.dirty_pr = false;
        }
    }
    // launch boundary: everything that only lives on chip goes back to its HBM slot
    __device__ __forceinline__ void flush(RegsT& X) {
        if (NV <= 0) return;
        if (X.dirty_qg || X.dirty_pr) { if (LEAN) lean_store(lean_rs(), X, X.dirty_qg, X.dirty_pr); else store_state(X, X.dirty_qg, X.dirty_pr); }
        if (c->phase == PH_TREE) {
            if (LEAN) {
                lean_ring_flush(lean_rs(), X);
            } else {
                const int64_t d = c->depth;
                if (X.ring_leaf0 >= 0) flush_ring_slot(0, first_slot_of(X.ring_leaf0, d));
                if (X.ring_leaf1 >= 0) flush_ring_slot(1, slot_last(__builtin_ctzll((unsigned long long)X.ring_leaf1), A.cap));
            }
        }
    }
    __device__ __forceinline__ void flush_ring_slot(int sl, int64_t slot) {
        constexpr int nk = NVX;  // the register kernels are instantiated per exact chunk count
        double2 tp[NVX], tr[NVX];
        ring_read(sl, tp, tr);
        double *pn = P(slot), *rn = R(slot);
#pragma unroll
        for (int k = 0; k < NVX; ++k) if (k < nk) { st2(pn, ridx(k), tp[k]); st2(rn, ridx(k), tr[k]); }
    }

    // ---- control state of a run of tree leaves, in registers.
    // With one wave per SIMD a dependent LDS round trip costs that wave ~100 cycles, and the control block lives in LDS: the leaf
    // loop used to read and write some forty of its words per leaf (leaf counter, cursor, pending leapfrog, collector sums ...),
    // and the Q-pool allocation walked the open sub-trees in a loop of dependent LDS reads.  During a run of leaves the words
    // every leaf touches live here instead — uniform values, i.e. SGPRs — together with two bit masks that make the allocation
    // three scalar instructions: the buffers pinned for the whole doubling (draw candidate, trajectory ends) and the buffers
    // held by open sub-trees (maintained as sub-trees are merged and opened).  What only changes when a doubling completes
    // (trajectory ends, candidate, main-tree weight, depth) stays in the control block.  Loaded when a run starts, written back
    // before anything else looks at the control block (rare paths, launch boundary).  Same arithmetic in the same order.
    struct Hot {
        int32_t nleaf, depth, dir, idx_cur;
        int32_t srcq, srcp, newq, newp;       // the pending leapfrog (in a tree: source = cursor, sign = dir)
        int32_t n_steps, n_steps0;            // (n_steps0: the value loaded; total_steps advances by the difference)
        int32_t end_newp;                     // P-slot of the leaf that completes the doubling
        uint32_t used_base, sub_used, draw;
        bool check;                           // U-turn criteria apply in this doubling
        double step, H0, acc, acc_sym, max_ee;
    };
    // ENDOUT: the leaf does not call the out-of-line end of a draw itself: it says how the draw ended, and run() makes the ONE call behind the loop of leaves,
    // where nothing of the leaf's state is alive (a call in the middle of the leaf pins what is alive there to the registers its callee leaves alone).
    static __device__ __forceinline__ int32_t end_code(bool diverging, bool maxdepth, bool store_div, bool div_has_end, bool regrad, bool replay) {
        return 1 | (diverging ? 2 : 0) | (maxdepth ? 4 : 0) | (store_div ? 8 : 0) | (div_has_end ? 16 : 0) | (regrad ? 32 : 0) | (replay ? 64 : 0);
    }
    static __device__ __forceinline__ int32_t rfl(int64_t v) { return __builtin_amdgcn_readfirstlane((int)v); }
    static __device__ __forceinline__ double rfl_f64(double v) {
        return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)), __builtin_amdgcn_readfirstlane(__double2loint(v)));
    }
    // what is fixed for the duration of a doubling (read from the control block when a run or a doubling starts)
    __device__ __forceinline__ void hot_doubling(Hot& H) const {
        const int db = H.dir > 0 ? 1 : 0;
        H.used_base = (1u << rfl(c->cand_q)) | (1u << rfl(c->endq[0])) | (1u << rfl(c->endq[1]));
        H.end_newp = slot_end(db, (int)(rfl(c->endpar[db]) ^ 1));
        H.check = A.s.check_turning && ((int64_t)H.depth + 1 > A.s.mindepth);
    }
    __device__ __forceinline__ void hot_load(Hot& H) const {
        H.nleaf = rfl(c->nleaf); H.depth = rfl(c->depth); H.dir = rfl(c->dir); H.idx_cur = rfl(c->idx_cur);
        H.srcq = rfl(c->lf_srcq); H.srcp = rfl(c->lf_srcp); H.newq = rfl(c->lf_newq); H.newp = rfl(c->lf_newp);
        H.n_steps = H.n_steps0 = rfl(c->n_steps); H.draw = (uint32_t)rfl(c->draw);
        H.step = rfl_f64(c->step_size); H.H0 = rfl_f64(c->H0); H.acc = rfl_f64(c->acc_sum); H.acc_sym = rfl_f64(c->acc_sym_sum);
        H.max_ee = A.s.max_energy_error;
        hot_doubling(H);
        H.sub_used = 0u;
        for (uint32_t m = (uint32_t)H.nleaf & ((1u << kMaxDepthCap) - 1u); m != 0; m &= m - 1) H.sub_used |= 1u << rfl(c->sub_q[__builtin_ctz(m)]);
    }
    __device__ __forceinline__ void hot_save(Hot& H) {
        c->nleaf = H.nleaf; c->dir = H.dir; c->idx_cur = H.idx_cur;
        c->lf_srcq = H.srcq; c->lf_srcp = H.srcp; c->lf_newq = H.newq; c->lf_newp = H.newp; c->lf_sign = H.dir; c->eval_buf = H.newq;
        c->curq = H.srcq; c->curp = H.srcp;
        c->n_steps = H.n_steps; c->total_steps += (int64_t)(H.n_steps - H.n_steps0);
        H.n_steps0 = H.n_steps;
        c->acc_sum = H.acc; c->acc_sym_sum = H.acc_sym;
    }
    // the next leaf of the doubling (issue_leaf): P-slot by leaf number; Q-pool buffer = the first one that is neither pinned
    // for the doubling, held by an open sub-tree nor the cursor
    __device__ __forceinline__ void issue_leaf_hot(Hot& H) const {
        const int32_t j = H.nleaf + 1, d = H.depth;
        int32_t newp;
        if (j == (1 << d)) newp = H.end_newp;
        else if (j & 1) newp = (j == 1) ? slot_first(d) : slot_first(__builtin_ctz((unsigned)(j - 1)));
        else newp = slot_last(__builtin_ctz((unsigned)j), A.cap);
        H.newq = (int32_t)__builtin_ctz(~(H.used_base | H.sub_used | (1u << H.srcq)));
        H.newp = newp;
    }
    // (the trajectory ends / depth of the finished doubling are in the control block by now)
    __device__ __forceinline__ void start_doubling_hot(Hot& H) const {
        nphip_u32x4 r = nphip_philox(A.s.seed, (uint32_t)H.depth, gchain, H.draw, NPHIP_RNG_DIRECTION);
        H.dir = (r.v[0] & 1u) ? 1 : -1;
        const int db = H.dir > 0 ? 1 : 0;
        H.nleaf = 0;
        H.srcq = rfl(c->endq[db]);
        H.srcp = rfl(c->endp[db]);
        H.idx_cur = rfl(H.dir > 0 ? c->idx_right : c->idx_left);
        H.sub_used = 0u;
        hot_doubling(H);
        issue_leaf_hot(H);
    }
    __device__ __forceinline__ double merge_uniform_hot(const Hot& H, int32_t j, int32_t d, int32_t k, nphip_u32x4& blk, int32_t& blk_id) const {
        const int32_t id = k >> 1;
        if (blk_id != id) {
            const uint32_t c3 = (uint32_t)NPHIP_RNG_MERGE | ((uint32_t)d << 8) | ((uint32_t)id << 16);
            blk = nphip_philox(A.s.seed, (uint32_t)j, gchain, H.draw, c3);
            blk_id = id;
        }
        return (k & 1) ? nphip_u01(blk.v[2], blk.v[3]) : nphip_u01(blk.v[0], blk.v[1]);
    }

    // multinomial merge of the waiting sub-tree of level k into T: keep T's draw w.p. w_T / (w_A + w_T)
    __device__ __forceinline__ void merge_level(Hot& H, int32_t j, int32_t d, int32_t k, nphip_u32x4& blk, int32_t& blk_id, double& T_wm, int64_t& T_we,
                                                int32_t& T_q, double& T_U, double& T_E, int32_t& T_idx) {
        double sm; int64_t se;
        // (the merged sub-tree's buffer is free again unless its draw survives as T's: the caller sets T's bit when it stores T)
        // (everything about the waiting sub-tree is read up front: one LDS round trip, not a second one behind the comparison)
        const int32_t A_q = rfl(c->sub_q[k]), A_idx = rfl(c->sub_idx[k]);
        const double A_U = c->sub_U[k], A_E = c->sub_E[k];
        H.sub_used &= ~(1u << A_q);
        nphip_w_add(c->sub_wm[k], c->sub_we[k], T_wm, T_we, &sm, &se);
        const bool take = merge_uniform_hot(H, j, d, k, blk, blk_id) * sm < nphip_w_rel(T_wm, T_we, se);
        if (!take) { T_q = A_q; T_U = A_U; T_E = A_E; T_idx = A_idx; }
        T_wm = sm; T_we = se;
    }

    // returns true when an out-of-line (rare) path ran
    __device__ __forceinline__ bool leaf_reg(RegsT& X, Hot& H, int32_t& end_code_out) {
        constexpr int nk = NVX;  // the register kernels are instantiated per exact chunk count
        const int32_t j = H.nleaf + 1, d = H.depth, dir = H.dir;
        const int db = dir > 0 ? 1 : 0;
        const int32_t idx_new = H.idx_cur + dir;
        const int32_t srcq = H.srcq, srcp = H.srcp, newq = H.newq, newp = H.newp;
        const bool check = H.check;
#ifdef NPHIP_PROFILE
        const int64_t tp0 = (int64_t)__builtin_readcyclecounter();
#endif
        // ---- source state (already in registers unless the cursor moved or a rare path ran)
        if (X.reg_q != srcq) {
            const double* q = Q(srcq);
#pragma unroll
            for (int k = 0; k < NVX; ++k) if (k < nk) X.q[k] = ld2(q, ridx(k));
            if (REMOTE) {   // the pool keeps (q, grad) of every stored position of a callback model
                const double* g = G(srcq);
#pragma unroll
                for (int k = 0; k < NVX; ++k) if (k < nk) X.g[k] = ld2(g, ridx(k));
            } else {
                regs_grad(X);
            }
        }
        if (X.reg_p != srcp) { load_slot(srcp, X.p, X.r); if (LR) load_slot_v(srcp, X.v); }
        if (!X.sig_ok) {
#pragma unroll
            for (int k = 0; k < NVX; ++k) if (k < nk) X.s[k] = ld2(sig2, ridx(k));
            if (LR) {
                const double* sdp = A.lr_std + (size_t)chain * ld;
#pragma unroll
                for (int k = 0; k < NVX; ++k) X.sd[k] = ld2(sdp, ridx(k));
            }
            X.sig_ok = true;
        }
        if (j == 1) { X.ring_leaf0 = -1; X.ring_leaf1 = -1; }
        // low-rank metric: does this chain integrate under a handed-in metric yet (else: the diagonal metric it adapts itself, with
        // the velocity sigma^2 p carried along so that every P-slot of the job has one)
        const bool lrm = LR && lr_job() && rfl(c->host_metric) != 0;
        const int lrk = lrm ? (int)rfl(c->lr_k) : 0;
        NPHIP_PHASE_FENCE();
        // ---- leapfrog (model parameters from LDS, neighbours by DPP, level-0 criterion in the same pass)
        const double eps = (double)dir * H.step;
        const double h = 0.5 * eps;
        if (idx_new == -1) {  // first backward step: rho' = p'  (-0.0 + p == p exactly, also for signed zeros)
#pragma unroll
            for (int k = 0; k < NVX; ++k) if (k < nk) { X.r[k].x = -0.0; X.r[k].y = -0.0; }
        }
        double2 z[NVX], pold[NVX], rold[NVX], vold[NVX];
        LrAcc lrS;
        double lrc[kLrMax];
        if (LR && lrm) lr_zero(lrS);
#pragma unroll
        for (int k = 0; k < NVX; ++k) if (k < nk) {
            pold[k] = X.p[k];
            rold[k] = X.r[k];
            if (LR) vold[k] = X.v[k];
            X.p[k].x = fma(h, X.g[k].x, X.p[k].x);
            X.p[k].y = fma(h, X.g[k].y, X.p[k].y);
            if (LR && lrm) {   // the products of u = std p_half with the columns first (below); the drift follows the reduction
            } else {
                X.q[k].x = fma(eps, X.s[k].x * X.p[k].x, X.q[k].x);
                X.q[k].y = fma(eps, X.s[k].y * X.p[k].y, X.q[k].y);
            }
            if (!REMOTE && !(LR && lrm)) {
                const double2 mu = par_mu(ridx(k));
                z[k].x = X.q[k].x - mu.x;
                z[k].y = X.q[k].y - mu.y;
            }
        }
        if (LR && lrm) {
            lr_dispatch(lrk, [&](auto kc) {
#pragma unroll
                for (int k = 0; k < NVX; ++k) {
                    double2 u;
                    u.x = X.sd[k].x * X.p[k].x; u.y = X.sd[k].y * X.p[k].y;
                    lr_acc_c<decltype(kc)::value>(lrS, ridx(k), u);
                }
            });
            lr_coef(lrS, lrk, 0, lrc);
            lr_dispatch(lrk, [&](auto kc) {
#pragma unroll
                for (int k = 0; k < NVX; ++k) {
                    double2 u;
                    u.x = X.sd[k].x * X.p[k].x; u.y = X.sd[k].y * X.p[k].y;
                    const double2 w = lr_apply_c<decltype(kc)::value>(ridx(k), u, lrc);
                    X.q[k].x = fma(eps, X.sd[k].x * w.x, X.q[k].x);
                    X.q[k].y = fma(eps, X.sd[k].y * w.y, X.q[k].y);
                }
            });
            if (!REMOTE) {
#pragma unroll
                for (int k = 0; k < NVX; ++k) {
                    const double2 mu = par_mu(ridx(k));
                    z[k].x = X.q[k].x - mu.x;
                    z[k].y = X.q[k].y - mu.y;
                }
            }
        }
        double lp_remote = 0.0;
        int64_t code_remote = 0;
        if (REMOTE) {
            // the evaluation is a remote call: position to the host's staging row, rendezvous, results back (remote_sync)
            if (DENS) {
#pragma unroll
                for (int k = 0; k < NVX; ++k) if (k < nk) *(NPHIP_LDS double2*)(dens_rows + ridx(k)) = X.q[k];
                remote_eval(lp_remote, code_remote, true);
            } else {
#pragma unroll
                for (int k = 0; k < NVX; ++k) if (k < nk) qe_st(ridx(k), X.q[k]);
                remote_eval(lp_remote, code_remote, false, !DG);
            }
        } else {
            publish_edges(z);
        }
        double2 accK = {0.0, 0.0}, accL = {0.0, 0.0}, accE = {0.0, 0.0}, accS = {0.0, 0.0};
        if (LR && lrm) lr_zero(lrS);
#pragma unroll
        for (int k = 0; k < NVX; ++k) if (k < nk) {
            double2 gg;
            if (DENS) {
                gg = *(const NPHIP_LDS double2*)(dens_rows + ld + ridx(k));
                if (ridx(k) >= D) gg.x = 0.0;         // (the density writes grad[0 .. dim) only)
                if (ridx(k) + 1 >= D) gg.y = 0.0;
            } else if (REMOTE) {
                gg = ge_ld(ridx(k));
                if (DG) {   // logp's sum rides along (z = x - mu; mu is padded with zeros to the row's length)
                    const double2 mu = ld2(A.dg_mu, ridx(k));
                    accL.x = fma(X.q[k].x - mu.x, gg.x, accL.x);
                    accL.y = fma(X.q[k].y - mu.y, gg.y, accL.y);
                }
            } else {
                double2 a, b01;          // b01 = b_{i-1}, b_i ; b2 = b_{i+1}
                double b2, edge_zl, edge_zr;
                par_ab(ridx(k), a, b01, b2);
                edge_pair(z, k, edge_zl, edge_zr);
                const double zl = wave_shr1(z[k].y, edge_zl);   // z_{i-1}
                const double zr = wave_shl1(z[k].x, edge_zr);   // z_{i+2}
                // boundary terms need no branches: b_{-1} and b_{D-1..} are stored as -0.0 and t + (-0.0) == t
                double tx = a.x * z[k].x;
                tx = fma(b01.x, zl, tx);
                tx = fma(b01.y, z[k].y, tx);
                double ty = a.y * z[k].y;
                ty = fma(b01.y, z[k].x, ty);
                ty = fma(b2, zr, ty);
                gg.x = -tx;
                gg.y = -ty;
                accL.x = fma(z[k].x, gg.x, accL.x);
                accL.y = fma(z[k].y, gg.y, accL.y);
            }
            X.g[k] = gg;
            X.p[k].x = fma(h, gg.x, X.p[k].x);
            X.p[k].y = fma(h, gg.y, X.p[k].y);
            X.r[k].x = rold[k].x + X.p[k].x;
            X.r[k].y = rold[k].y + X.p[k].y;
            if (LR && lrm) continue;   // the velocity of the new state needs the products with the columns first: below
            const double vx = X.s[k].x * X.p[k].x, vy = X.s[k].y * X.p[k].y;
            accK.x = fma(X.p[k].x, vx, accK.x);
            accK.y = fma(X.p[k].y, vy, accK.y);
            // level-0 criterion between source and new leaf: span = (rho' - rho) + p
            const double tx0 = (X.r[k].x - rold[k].x) + pold[k].x, ty0 = (X.r[k].y - rold[k].y) + pold[k].y;
            accE.x = fma(tx0, vx, accE.x);
            accE.y = fma(ty0, vy, accE.y);
            if (LR) {   // (a job that may receive metrics: every state carries its velocity, here sigma^2 p)
                X.v[k].x = vx; X.v[k].y = vy;
                accS.x = fma(tx0, vold[k].x, accS.x);
                accS.y = fma(ty0, vold[k].y, accS.y);
            } else {
                accS.x = fma(tx0, X.s[k].x * pold[k].x, accS.x);
                accS.y = fma(ty0, X.s[k].y * pold[k].y, accS.y);
            }
        }
        if (LR && lrm) {
            lr_dispatch(lrk, [&](auto kc) {
#pragma unroll
                for (int k = 0; k < NVX; ++k) {
                    double2 u;
                    u.x = X.sd[k].x * X.p[k].x; u.y = X.sd[k].y * X.p[k].y;
                    lr_acc_c<decltype(kc)::value>(lrS, ridx(k), u);
                }
            });
            lr_coef(lrS, lrk, 0, lrc);
            lr_dispatch(lrk, [&](auto kc) {
#pragma unroll
                for (int k = 0; k < NVX; ++k) {
                    double2 u;
                    u.x = X.sd[k].x * X.p[k].x; u.y = X.sd[k].y * X.p[k].y;
                    const double2 w = lr_apply_c<decltype(kc)::value>(ridx(k), u, lrc);
                    X.v[k].x = X.sd[k].x * w.x; X.v[k].y = X.sd[k].y * w.y;
                }
            });
#pragma unroll
            for (int k = 0; k < NVX; ++k) {
                accK.x = fma(X.p[k].x, X.v[k].x, accK.x);
                accK.y = fma(X.p[k].y, X.v[k].y, accK.y);
                const double tx0 = (X.r[k].x - rold[k].x) + pold[k].x, ty0 = (X.r[k].y - rold[k].y) + pold[k].y;
                accE.x = fma(tx0, X.v[k].x, accE.x);
                accE.y = fma(ty0, X.v[k].y, accE.y);
                accS.x = fma(tx0, vold[k].x, accS.x);
                accS.y = fma(ty0, vold[k].y, accS.y);
            }
        }
        X.reg_q = newq;
        X.reg_p = newp;
        X.dirty_qg = true;
        X.dirty_pr = true;
        // (Measured and rejected in round 6, profiles/r6_headline_merge_operand_prefetch_rejected.txt: fetching the operands of the level >= 2 merges this
        // leaf closes ahead of their use, from here — into 128 accumulation registers (a vector load can target them on gfx950): the allocator's own spills
        // overflow into scratch, 5 x slower; one touch per cache line into one register, or into LDS by the memory-to-LDS loads (no register at all): the
        // merges' loads then hit L2 and the level >= 1 checks cost 125 cycles per leaf less, of 2 k — they are issue, not latency — while the control flow
        // added to this block costs the leapfrog's schedule 240: 209 against 217 M leapfrogs/s.)
        // the two most recent summaries a level-1 merge needs stay on chip
        if (!LR && !NORING) {
            if ((j & 3) == 1) { ring_write(0, X.p, X.r); X.ring_leaf0 = j; }
            else if ((j & 3) == 2) { ring_write(1, X.p, X.r); X.ring_leaf1 = j; }
        }
#ifdef NPHIP_PROFILE
        const int64_t tp1 = (int64_t)__builtin_readcyclecounter();
#endif
        NPHIP_PHASE_FENCE();
        double v4[4] = {accK.x + accK.y, accL.x + accL.y, accE.x + accE.y, accS.x + accS.y};
        rsum(v4);
        NPHIP_PHASE_FENCE();
#ifdef NPHIP_PROFILE
        const int64_t tp2 = (int64_t)__builtin_readcyclecounter();
        c->prof[0] += tp1 - tp0; c->prof[6] += tp2 - tp1;
#endif
        const double K = 0.5 * v4[0], lp = (REMOTE && !DG) ? lp_remote : 0.5 * v4[1];
        const bool turn0 = (v4[2] < 0.0) || (v4[3] < 0.0);
        if (REMOTE && code_remote < 0) { X.dirty_qg = X.dirty_pr = false; hot_save(H); finish_chain(PH_ERROR, CE_FATAL_LOGP); return true; }
        // ---- NutsTree::extend / merge_into, unrolled (same decisions as cont_tree)
        H.nleaf += 1;
        H.n_steps += 1;
        const bool ok = isfinite(lp) && (!REMOTE || code_remote == 0);
        const double Unew = -lp, E = K + Unew, dE = E - H.H0;
        const bool diverged = !ok || (dE > H.max_ee) || !isfinite(dE);
        double T_wm = 1.0;
        int64_t T_we = 0;
        nphip_u32x4 mrg_blk = {{0u, 0u, 0u, 0u}};
        int32_t mrg_id = -1;
        {
            if (!diverged) {
                // one exp serves the collector and the leaf's multinomial weight (its (p, k) parts)
                const double x = -dE, xc = x > 1e9 ? 1e9 : (x < -1e9 ? -1e9 : x);
                double kk;
                nphip_exp_parts(xc, &T_wm, &kk);
                T_we = (int64_t)kk;
                const double e = nphip_exp_scale(x, T_wm, kk);
                const double a = e < 1.0 ? e : 1.0;
                H.acc += a;
                H.acc_sym += 2.0 * a / (1.0 + e);
            }
        }
        if (diverged) { X.dirty_qg = X.dirty_pr = false; hot_save(H);
            if (ENDOUT) end_code_out = end_code(true, false, FUSED, ok, !REMOTE, FUSED); else rare_end_draw(A, c, red, chain, true, false, FUSED, ok, !REMOTE, FUSED);
            return true; }
#ifdef NPHIP_PROFILE
        int64_t tq = (int64_t)__builtin_readcyclecounter();
        c->prof[8] += tq - tp2;
#endif

        NPHIP_PHASE_FENCE();
        double T_U = Unew, T_E = E;
        int32_t T_q = newq, T_idx = idx_new;
        H.srcq = newq; H.srcp = newp; H.idx_cur = idx_new;   // the cursor moves to the new leaf
        double2 obp[NVX], obr[NVX], obv[NVX];  // one operand buffer (A.first, then A.last of the merge being checked)
        int32_t k = 0;
        while (k < d && (((j - 1) >> k) & 1)) {
            if (check) {
                bool turn;
                if (k == 0) {
                    turn = turn0;
                } else if (LR) {
                    // low-rank metric: no LDS ring — every summary comes from its P-slot (p, rho, v); T.first as in the other branch
                    const int32_t a = j - (2 << k) + 1, al = j - (1 << k);
                    load_slot(first_slot_of(a, d), obp, obr); load_slot_v(first_slot_of(a, d), obv);
                    if (k == 2) { load_slot(first_slot_of(al + 1, d), pold, rold); load_slot_v(first_slot_of(al + 1, d), vold); }
                    turn = sub_a_v(X, obp, obr, obv, pold, rold, vold);
                    if (k >= 2) {
#pragma unroll
                        for (int q_ = 0; q_ < NVX; ++q_) { pold[q_] = obp[q_]; rold[q_] = obr[q_]; vold[q_] = obv[q_]; }
                    }
                    if (!turn) {
                        const int32_t sl_ = slot_last(__builtin_ctz((unsigned)al), A.cap);
                        load_slot(sl_, obp, obr); load_slot_v(sl_, obv);
                        turn = sub_b_v(X, obp, obr, obv);
                    }
                } else {
                    const int32_t a = j - (2 << k) + 1, al = j - (1 << k);
                    // A.first
                    if (k == 1 && X.ring_leaf0 == a) ring_read(0, obp, obr);
                    else load_slot(first_slot_of(a, d), obp, obr);
                    // T.first: level 1 -> source registers (pold, rold); level 2 -> leaf j-3 (ring slot 0);
                    // level >= 3 -> A.first of the level below, moved into (pold, rold) there
                    if (k == 2) {
                        if (X.ring_leaf0 == al + 1) ring_read(0, pold, rold);
                        else load_slot(first_slot_of(al + 1, d), pold, rold);
                    }
                    turn = sub_a(X, obp, obr, pold, rold);
                    if (k >= 2) {
#pragma unroll
                        for (int q_ = 0; q_ < NVX; ++q_) { pold[q_] = obp[q_]; rold[q_] = obr[q_]; }
                    }
                    if (!turn) {
                        if (k == 1 && X.ring_leaf1 == al) ring_read(1, obp, obr);
                        else load_slot(slot_last(__builtin_ctz((unsigned)al), A.cap), obp, obr);
                        turn = sub_b(X, obp, obr);
                    }
                }
                if (turn) { X.dirty_qg = X.dirty_pr = false; hot_save(H); if (ENDOUT) end_code_out = end_code(false, false, false, false, !REMOTE, false); else rare_end_draw(A, c, red, chain, false, false, false, false, !REMOTE); return true; }
            }
#ifdef NPHIP_PROFILE
            { const int64_t t_ = (int64_t)__builtin_readcyclecounter(); c->prof[k == 0 ? 9 : 10] += t_ - tq; tq = t_; }
#endif
            NPHIP_PHASE_FENCE();
            merge_level(H, j, d, k, mrg_blk, mrg_id, T_wm, T_we, T_q, T_U, T_E, T_idx);
            NPHIP_PHASE_FENCE();
#ifdef NPHIP_PROFILE
            { const int64_t t_ = (int64_t)__builtin_readcyclecounter(); c->prof[11] += t_ - tq; tq = t_; }
#endif
            ++k;
        }
        if (k < d) {
            c->sub_wm[k] = T_wm; c->sub_we[k] = T_we; c->sub_q[k] = T_q; c->sub_U[k] = T_U; c->sub_E[k] = T_E; c->sub_idx[k] = T_idx;
            H.sub_used |= 1u << T_q;
            // ---- stores, last: q when the leaf is referenced as a candidate; (p, rho) only for leaves a level >= 2
            // merge reads back from HBM (leaf % 4 == 0: A.last, leaf % 8 == 1: A.first)
#ifdef NPHIP_PROFILE
            const int64_t tp3 = (int64_t)__builtin_readcyclecounter();
#endif
            store_state(X, T_q == newq, LR || NORING || ((j & 3) == 0) || ((j & 7) == 1));   // (low-rank, wide rows: no ring, every summary goes to its slot)
            issue_leaf_hot(H);
#ifdef NPHIP_PROFILE
            c->prof[7] += (int64_t)__builtin_readcyclecounter() - tp3;
#endif
            return false;
        }
        // ---- the new sub-tree of depth d is complete (j == 2^d): merge into the main tree (general index modes)
        bool turn = false;
        if (check) {
            const int32_t far_slot = rfl(c->endp[1 - db]), far_idx = rfl(dir > 0 ? c->idx_left : c->idx_right);
            const int32_t near_idx = rfl(dir > 0 ? c->idx_right : c->idx_left);
            if (d == 0) {
                // both ends are the origin = the source of this leapfrog, and rho_0 == p_0
                turn = LR ? check1_v(X, pold, pold, vold, far_idx, idx_new) : check1(X, pold, pold, far_idx, idx_new);
            } else if (LR) {
                if (d == 2) { load_slot(slot_first((int)d), pold, rold); load_slot_v(slot_first((int)d), vold); }
                load_slot(far_slot, obp, obr); load_slot_v(far_slot, obv);
                turn = check_a_v(X, obp, obr, obv, pold, rold, vold, far_idx, near_idx + dir, idx_new);
                if (!turn) {
                    load_slot(rfl(c->endp[db]), obp, obr); load_slot_v(rfl(c->endp[db]), obv);
                    turn = check1_v(X, obp, obr, obv, near_idx, idx_new);
                }
            } else {
                // T.first = leaf 1: d == 1 -> source registers; d == 2 -> ring slot 0; d >= 3 -> already in (pold, rold)
                if (d == 2) {
                    if (X.ring_leaf0 == 1) ring_read(0, pold, rold);
                    else load_slot(slot_first((int)d), pold, rold);
                }
                load_slot(far_slot, obp, obr);
                turn = check_a(X, obp, obr, pold, rold, far_idx, near_idx + dir, idx_new);   // (far, TL) || (far, TF)
                if (!turn) {
                    load_slot(rfl(c->endp[db]), obp, obr);
                    turn = check1(X, obp, obr, near_idx, idx_new);                           // (near, TL)
                }
            }
        }
        c->endq[db] = newq;
        c->endp[db] = newp;
        c->endpar[db] ^= 1;
        if (dir > 0) c->idx_right = idx_new; else c->idx_left = idx_new;
        {
            // biased progressive sampling at the top level: take T's draw w.p. min(1, w_T / w_main)
            double sm; int64_t se;
            nphip_w_add(c->main_wm, c->main_we, T_wm, T_we, &sm, &se);
            const double ref = nphip_w_rel(c->main_wm, c->main_we, se), oth = nphip_w_rel(T_wm, T_we, se);
            bool take = oth >= ref;
            if (!take) take = merge_uniform_hot(H, j, d, d, mrg_blk, mrg_id) * ref < oth;
            if (take) { c->cand_q = T_q; c->cand_U = T_U; c->cand_E = T_E; c->cand_idx = T_idx; }
            c->main_wm = sm; c->main_we = se;
            H.depth = d + 1;
            c->depth = d + 1;
        }
        store_state(X, true, true);  // a new trajectory end is always written back
        if (turn) { hot_save(H); if (ENDOUT) end_code_out = end_code(false, false, false, false, !REMOTE, false); else rare_end_draw(A, c, red, chain, false, false, false, false, !REMOTE); return true; }
        if (H.depth >= A.s.maxdepth) { hot_save(H); if (ENDOUT) end_code_out = end_code(false, true, false, false, !REMOTE, false); else rare_end_draw(A, c, red, chain, false, true, false, false, !REMOTE); return true; }
        start_doubling_hot(H);
        return false;
    }

    // ======================================================================================
    // Lean register-resident leaf (LEAN): the design for long rows (one chain per CU at D = 10 000).
    //
    // What differs from leaf_reg: there is no LDS ring and no (pold, rold, operand) whole-vector temporaries — a
    // wave's registers hold its NV chunks of (q, grad, p, rho) and nothing else that scales with the row:
    //   * the leapfrog runs in two sweeps over the chunks (position update + edge publication, barrier, gradient +
    //     second half-kick); the first half-kick is recomputed in the second sweep (bit-identical) instead of kept;
    //   * the level-0 criterion is accumulated in the second sweep from the before/after values of each chunk;
    //   * the criteria of a level >= 1 merge stream their operands (A.first, A.last, T.first) from the P-slots chunk by
    //     chunk against the resident new leaf T.last;  (A.first, T.first) of a level-1 merge — whose T.first is the
    //     source of the NEXT leapfrog and therefore never written to HBM — is evaluated one leaf early, while that
    //     T.first is the resident leaf (leaves = 3 mod 4), and kept as one bit in the control block (pre_turn).
    // Decisions and floats are those of every other kernel: each criterion is its own pair of accumulators, summed per
    // (lane, component) over the wave's chunks in increasing order, then in the contract's reduction order.
    // ======================================================================================
    // Addressing of the lean kernels: buffer instructions — a 128-bit descriptor in SGPRs per array (this chain's Q-pool and
    // P-slots, the three model vectors), ONE per-lane 32-bit offset (lane * 16) shared by every access, and the (slot, vector,
    // chunk) position as a scalar offset.  With 64-bit per-lane global addresses the compiler hoists one address pair per
    // (array, chunk) out of the leaf loop and spills them all (measured: 1000 spilled VGPRs at 10 chunks per wave).
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
    struct LeanRs {
        __amdgpu_buffer_rsrc_t q, p, mu, a, b;
        uint32_t voff;      // lane * 16
        uint32_t wave_off;  // byte offset of this wave's first chunk inside a vector
    };
    static __device__ __forceinline__ __amdgpu_buffer_rsrc_t mk_rsrc(const void* ptr, uint64_t bytes) {
        const uint64_t v = (uint64_t)ptr;
        const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v), hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32));
        return __builtin_amdgcn_make_buffer_rsrc((void*)(((uint64_t)hi << 32) | lo), 0, (int)__builtin_amdgcn_readfirstlane((int)(uint32_t)bytes), 0x00020000);
    }
    __device__ __forceinline__ LeanRs lean_rs() const {
        LeanRs r;
        r.q = mk_rsrc(qp, (uint64_t)A.nqpool * 2 * ld * 8);
        r.p = mk_rsrc(pp, (uint64_t)A.npslots * 2 * ld * 8);
        r.mu = mk_rsrc(A.m_mu, (uint64_t)ld * 8);
        r.a = mk_rsrc(A.m_a, (uint64_t)ld * 8);
        r.b = mk_rsrc(A.m_bsh, (uint64_t)(ld + 8) * 8);
        r.voff = (uint32_t)lane * 16u;
        r.wave_off = (uint32_t)wave * (NPHIP_CHUNK * 8);
        return r;
    }
    // scalar byte offset of chunk k of this wave inside vector `vec` (0/1) of buffer / slot `slot`
    __device__ __forceinline__ uint32_t soff(const LeanRs& rs, int64_t slot, int vec, int k) const {
        const uint32_t s_ = (uint32_t)__builtin_amdgcn_readfirstlane((int)slot);
        return (s_ * 2u + (uint32_t)vec) * (uint32_t)(ld * 8) + (uint32_t)k * (uint32_t)(W * NPHIP_CHUNK * 8) + rs.wave_off;
    }
    static __device__ __forceinline__ double2 bld2(__amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t so) {
        return __builtin_bit_cast(double2, __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)so, 0));
    }
    static __device__ __forceinline__ void bst2(__amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t so, double2 v) {
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, (int)voff, (int)so, 0);
    }
    // model vectors: chunk k of this wave
    __device__ __forceinline__ double2 par2(const LeanRs& rs, __amdgpu_buffer_rsrc_t r, int k) const {
        return bld2(r, rs.voff, (uint32_t)k * (uint32_t)(W * NPHIP_CHUNK * 8) + rs.wave_off);
    }
    __device__ __forceinline__ double par_b2(const LeanRs& rs, int k) const {   // b_{i+1} = m_bsh[i + 2]
        return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rs.b, (int)(rs.voff + 16u), (int)((uint32_t)k * (uint32_t)(W * NPHIP_CHUNK * 8) + rs.wave_off), 0));
    }
    __device__ __forceinline__ double2 sigl(const LeanRs& rs, int k) const {
        return *(const NPHIP_LDS double2*)((const NPHIP_LDS char*)sig_lds + ((uint32_t)k * (uint32_t)(W * NPHIP_CHUNK * 8) + rs.wave_off + rs.voff));
    }

    // One (p, rho) summary on chip, as far as it fits beside sigma^2: the leaf = 2 mod 4 of a level-1 merge — A.last, which
    // nothing else ever reads — goes to LDS instead of HBM and is read back from there two leaves later.  Up to ld = 6144 the
    // whole summary fits (3 x 48 KB); above, the first KP chunks of p and KR chunks of rho of every wave do (D = 10 000: 18 of
    // the 20 chunks of p) and the rest goes through its P-slot as before.  Every wave touches only its own chunks: no barrier.
    // The LDS part is written back at a launch boundary (flush).
    static constexpr int LR_FREE = (LEAN && W == 4) ? (163840 - (NVX > 20 ? 12288 : 8192) - NVX * W * (NPHIP_CHUNK * 8)) / (W * (NPHIP_CHUNK * 8)) : 0;   // chunks per wave (the kernel's static LDS grows with the edge buffer: 8.1 KB at 20 chunks)
    static constexpr int KP = LR_FREE < 0 ? 0 : (LR_FREE < NVX ? LR_FREE : NVX);
    static constexpr int KR = (LR_FREE - KP) < 0 ? 0 : ((LR_FREE - KP) < NVX ? (LR_FREE - KP) : NVX);
    static constexpr bool LRING = KP > 0;
    __device__ __forceinline__ NPHIP_LDS double2* lring(const LeanRs& rs, int vec, int k) const {
        return (NPHIP_LDS double2*)((NPHIP_LDS char*)sig_lds + (uint32_t)(ld * 8) + (uint32_t)((vec == 0 ? 0 : KP) + k) * (uint32_t)(W * NPHIP_CHUNK * 8) +
                                    rs.wave_off + rs.voff);
    }
    __device__ __forceinline__ void lean_ring_flush(const LeanRs& rs, RegsT& X) {
        if (!LRING || X.ring_leaf1 < 0) return;
        const int64_t slot = slot_last(__builtin_ctzll((unsigned long long)X.ring_leaf1), A.cap);
#pragma unroll
        for (int k = 0; k < NVX; ++k) {
            if (k < KP) bst2(rs.p, rs.voff, soff(rs, slot, 0, k), *lring(rs, 0, k));
            if (k < KR) bst2(rs.p, rs.voff, soff(rs, slot, 1, k), *lring(rs, 1, k));
        }
        X.ring_leaf1 = -1;
    }

    // gradient of the resident position (after a reload: only q is kept in HBM) — two sweeps, edges through LDS
    __device__ __forceinline__ void lean_grad(const LeanRs& rs, RegsT& X) {
#pragma unroll
        for (int k = 0; k < NVX; ++k) {
            const double2 mu = par2(rs, rs.mu, k);
            const int64_t cch = (int64_t)k * W + wave;
            const double zx = X.q[k].x - mu.x, zy = X.q[k].y - mu.y;
            if (lane == 0 || lane == 63) edge[2 * cch + (lane == 0 ? 1 : 2)] = (lane == 0) ? zx : zy;
            NPHIP_CHUNK_FENCE(k);
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < NVX; ++k) {
            const int64_t cch = (int64_t)k * W + wave;
            const double2 mu = par2(rs, rs.mu, k), a = par2(rs, rs.a, k), b01 = par2(rs, rs.b, k);
            const double b2 = par_b2(rs, k);
            double2 z;
            z.x = X.q[k].x - mu.x;
            z.y = X.q[k].y - mu.y;
            const double ezl = edge[2 * cch], ezr = edge[2 * cch + 3];   // (the buffer is padded with 0.0 at both ends)
            const double zl = wave_shr1(z.y, ezl), zr = wave_shl1(z.x, ezr);
            double tx = a.x * z.x;
            tx = fma(b01.x, zl, tx);
            tx = fma(b01.y, z.y, tx);
            double ty = a.y * z.y;
            ty = fma(b01.y, z.x, ty);
            ty = fma(b2, zr, ty);
            X.g[k].x = -tx;
            X.g[k].y = -ty;
            NPHIP_CHUNK_FENCE(k);
        }
        __syncthreads();  // the edge buffer is free again
    }

    // ---- streamed criteria.  PF chunks of look-ahead: the operands of chunk k + PF are requested before chunk k is used.  A
    // pass is a pure memory phase of the whole CU (one chain per CU: every wave is in it at once), so what bounds it is the
    // number of bytes in flight: 4 waves with the state parked in AGPRs have VGPRs for 4 chunks x 3 slots, 8 waves for one.
    // (18+ chunks per wave: with four chunks in flight the passes spill 32-64 VGPRs, and a lone wave waits out every reload;
    //  measured, same box, 512 leapfrogs per launch: D = 9000 16.2 -> 16.9, 9700 13.8 -> 15.0, 10 000 12.5 -> 14.1 M leapfrogs/s with two;
    //  one or three in flight at 20 chunks: 12.4 / 12.2)
    static constexpr int PF = (W <= 4) ? (NV >= 18 ? 2 : 4) : 1;
    struct SlotChunk { double2 p, r; };
    __device__ __forceinline__ SlotChunk ldslot(const LeanRs& rs, int64_t slot, int k) const {
        SlotChunk c_;
        c_.p = bld2(rs.p, rs.voff, soff(rs, slot, 0, k));
        c_.r = bld2(rs.p, rs.voff, soff(rs, slot, 1, k));
        return c_;
    }
    // (A, resident) inside a doubling: A earlier, the resident leaf later
    __device__ __forceinline__ bool lean_pass1(const LeanRs& rs, const RegsT& X, int64_t sAf) {
        double2 e = {0.0, 0.0}, st = {0.0, 0.0};
        SlotChunk an[PF];
#pragma unroll
        for (int u = 0; u < PF; ++u) if (u < NVX) an[u] = ldslot(rs, sAf, u);
#pragma unroll
        for (int k = 0; k < NVX; ++k) {
            const SlotChunk a = an[k % PF];
            if (k + PF < NVX) an[k % PF] = ldslot(rs, sAf, k + PF);
            const double2 s2 = sigl(rs, k);
            span_acc(a.p.x, a.r.x, X.p[k].x, X.r[k].x, s2.x, e.x, st.x);
            span_acc(a.p.y, a.r.y, X.p[k].y, X.r[k].y, s2.y, e.y, st.y);
            NPHIP_CHUNK_FENCE(k);
        }
        double v[2] = {e.x + e.y, st.x + st.y};
        rsum(v);
        return (v[0] < 0.0) || (v[1] < 0.0);
    }
    // (A.first, resident) || (A.last, resident)
    // (RING: A.last comes from the LDS slot)
    template <bool RING>
    __device__ __forceinline__ bool lean_pass2(const LeanRs& rs, const RegsT& X, int64_t sAf, int64_t sAl) {
        double2 acc[4];
#pragma unroll
        for (int n = 0; n < 4; ++n) { acc[n].x = 0.0; acc[n].y = 0.0; }
        constexpr int kp = RING ? KP : 0, kr = RING ? KR : 0;   // chunks of A.last's p / rho that live in LDS
        SlotChunk an[PF], bn[PF];
#pragma unroll
        for (int u = 0; u < PF; ++u) if (u < NVX) {
            an[u] = ldslot(rs, sAf, u);
            if (u >= kp) bn[u].p = bld2(rs.p, rs.voff, soff(rs, sAl, 0, u));
            if (u >= kr) bn[u].r = bld2(rs.p, rs.voff, soff(rs, sAl, 1, u));
        }
#pragma unroll
        for (int k = 0; k < NVX; ++k) {
            const SlotChunk a = an[k % PF];
            SlotChunk b;
            if (k < kp) b.p = *lring(rs, 0, k); else b.p = bn[k % PF].p;
            if (k < kr) b.r = *lring(rs, 1, k); else b.r = bn[k % PF].r;
            if (k + PF < NVX) {
                an[k % PF] = ldslot(rs, sAf, k + PF);
                if (k + PF >= kp) bn[k % PF].p = bld2(rs.p, rs.voff, soff(rs, sAl, 0, k + PF));
                if (k + PF >= kr) bn[k % PF].r = bld2(rs.p, rs.voff, soff(rs, sAl, 1, k + PF));
            }
            const double2 s2 = sigl(rs, k);
            span_acc(a.p.x, a.r.x, X.p[k].x, X.r[k].x, s2.x, acc[0].x, acc[1].x);
            span_acc(a.p.y, a.r.y, X.p[k].y, X.r[k].y, s2.y, acc[0].y, acc[1].y);
            span_acc(b.p.x, b.r.x, X.p[k].x, X.r[k].x, s2.x, acc[2].x, acc[3].x);
            span_acc(b.p.y, b.r.y, X.p[k].y, X.r[k].y, s2.y, acc[2].y, acc[3].y);
            NPHIP_CHUNK_FENCE(k);
        }
        double v[4];
#pragma unroll
        for (int n = 0; n < 4; ++n) v[n] = acc[n].x + acc[n].y;
        rsum(v);
        return (v[0] < 0.0) || (v[1] < 0.0) || (v[2] < 0.0) || (v[3] < 0.0);
    }
    // (A.first, resident) || (A.first, T.first) || (A.last, resident)
    __device__ __forceinline__ bool lean_pass3(const LeanRs& rs, const RegsT& X, int64_t sAf, int64_t sAl, int64_t sTf) {
        double2 acc[6];
#pragma unroll
        for (int n = 0; n < 6; ++n) { acc[n].x = 0.0; acc[n].y = 0.0; }
        SlotChunk an[PF], bn[PF], fn[PF];
#pragma unroll
        for (int u = 0; u < PF; ++u) if (u < NVX) { an[u] = ldslot(rs, sAf, u); bn[u] = ldslot(rs, sAl, u); fn[u] = ldslot(rs, sTf, u); }
#pragma unroll
        for (int k = 0; k < NVX; ++k) {
            const SlotChunk a = an[k % PF], b = bn[k % PF], f = fn[k % PF];
            if (k + PF < NVX) { an[k % PF] = ldslot(rs, sAf, k + PF); bn[k % PF] = ldslot(rs, sAl, k + PF); fn[k % PF] = ldslot(rs, sTf, k + PF); }
            const double2 s2 = sigl(rs, k);
            span_acc(a.p.x, a.r.x, X.p[k].x, X.r[k].x, s2.x, acc[0].x, acc[1].x);
            span_acc(a.p.y, a.r.y, X.p[k].y, X.r[k].y, s2.y, acc[0].y, acc[1].y);
            span_acc(a.p.x, a.r.x, f.p.x, f.r.x, s2.x, acc[2].x, acc[3].x);
            span_acc(a.p.y, a.r.y, f.p.y, f.r.y, s2.y, acc[2].y, acc[3].y);
            span_acc(b.p.x, b.r.x, X.p[k].x, X.r[k].x, s2.x, acc[4].x, acc[5].x);
            span_acc(b.p.y, b.r.y, X.p[k].y, X.r[k].y, s2.y, acc[4].y, acc[5].y);
            NPHIP_CHUNK_FENCE(k);
        }
        double v[6];
#pragma unroll
        for (int n = 0; n < 6; ++n) v[n] = acc[n].x + acc[n].y;
        rsum(v);
        return (v[0] < 0.0) || (v[1] < 0.0) || (v[2] < 0.0) || (v[3] < 0.0) || (v[4] < 0.0) || (v[5] < 0.0);
    }
    // General index modes without branches: the span of every (first_is_start, mode) combination is
    //   (U - V) + Wv  with  v0: (r2 - r1) + p1   v1: (r1 - r2) + p2   v2: r2 - (-r1) + (-0.0) = r1 + r2  (exact: x + (-0.0) == x),
    // selected by wave-uniform flags.  acc2 takes span . sigma^2 p2, acc1 span . sigma^2 p1 (which of the two is "end" and
    // which "start" only names them: both are tested against zero).
    struct PairU { bool swap, sum; };
    __device__ __forceinline__ PairU pair_u(const Pair pr) const {
        PairU u;
        u.sum = pr.mode == 1;
        u.swap = pr.first_is_start ? (pr.mode == 2) : (pr.mode == 0);
        return u;
    }
    __device__ __forceinline__ void pair_acc_u(const PairU u, double p1, double r1, double p2, double r2, double s2v, double& acc2, double& acc1) const {
        const double U = u.swap ? r1 : r2;
        double V = u.swap ? r2 : r1;
        V = u.sum ? -V : V;
        const double Wv = u.sum ? -0.0 : (u.swap ? p2 : p1);
        const double t = (U - V) + Wv;
        acc2 = fma(t, s2v * p2, acc2);
        acc1 = fma(t, s2v * p1, acc1);
    }
    // top-level merge (general index modes): (far, resident) q1 || (far, T.first) q3 || (near, resident) qn;
    // d == 0 (full == false): only (far, resident) — both ends are the origin
    __device__ __forceinline__ bool lean_top(const LeanRs& rs, const RegsT& X, bool full, int64_t sFar, int64_t sNear, int64_t sTf, const Pair q1,
                                             const Pair q3, const Pair qn) {
        const PairU p1 = pair_u(q1), p3 = pair_u(q3), pn = pair_u(qn);
        double2 acc[6];
#pragma unroll
        for (int n = 0; n < 6; ++n) { acc[n].x = 0.0; acc[n].y = 0.0; }
        if (full) {
            SlotChunk an[PF], bn[PF], fn[PF];
#pragma unroll
            for (int u = 0; u < PF; ++u) if (u < NVX) { an[u] = ldslot(rs, sFar, u); bn[u] = ldslot(rs, sNear, u); fn[u] = ldslot(rs, sTf, u); }
#pragma unroll
            for (int k = 0; k < NVX; ++k) {
                const SlotChunk a = an[k % PF], b = bn[k % PF], f = fn[k % PF];
                if (k + PF < NVX) { an[k % PF] = ldslot(rs, sFar, k + PF); bn[k % PF] = ldslot(rs, sNear, k + PF); fn[k % PF] = ldslot(rs, sTf, k + PF); }
                const double2 s2 = sigl(rs, k);
                pair_acc_u(p1, a.p.x, a.r.x, X.p[k].x, X.r[k].x, s2.x, acc[0].x, acc[1].x);
                pair_acc_u(p1, a.p.y, a.r.y, X.p[k].y, X.r[k].y, s2.y, acc[0].y, acc[1].y);
                pair_acc_u(p3, a.p.x, a.r.x, f.p.x, f.r.x, s2.x, acc[2].x, acc[3].x);
                pair_acc_u(p3, a.p.y, a.r.y, f.p.y, f.r.y, s2.y, acc[2].y, acc[3].y);
                pair_acc_u(pn, b.p.x, b.r.x, X.p[k].x, X.r[k].x, s2.x, acc[4].x, acc[5].x);
                pair_acc_u(pn, b.p.y, b.r.y, X.p[k].y, X.r[k].y, s2.y, acc[4].y, acc[5].y);
                NPHIP_CHUNK_FENCE(k);
            }
        } else {
            SlotChunk an[PF];
#pragma unroll
            for (int u = 0; u < PF; ++u) if (u < NVX) an[u] = ldslot(rs, sFar, u);
#pragma unroll
            for (int k = 0; k < NVX; ++k) {
                const SlotChunk a = an[k % PF];
                if (k + PF < NVX) an[k % PF] = ldslot(rs, sFar, k + PF);
                const double2 s2 = sigl(rs, k);
                pair_acc_u(p1, a.p.x, a.r.x, X.p[k].x, X.r[k].x, s2.x, acc[0].x, acc[1].x);
                pair_acc_u(p1, a.p.y, a.r.y, X.p[k].y, X.r[k].y, s2.y, acc[0].y, acc[1].y);
                NPHIP_CHUNK_FENCE(k);
            }
        }
        double v[6];
#pragma unroll
        for (int n = 0; n < 6; ++n) v[n] = acc[n].x + acc[n].y;
        rsum(v);
        return (v[0] < 0.0) || (v[1] < 0.0) || (v[2] < 0.0) || (v[3] < 0.0) || (v[4] < 0.0) || (v[5] < 0.0);
    }

    __device__ __forceinline__ void lean_store(const LeanRs& rs, RegsT& X, bool q_, bool pr) {
        if (q_) {
#pragma unroll
            for (int k = 0; k < NVX; ++k) bst2(rs.q, rs.voff, soff(rs, X.reg_q, 0, k), X.q[k]);
            X.dirty_qg = false;
        }
        if (pr) {
#pragma unroll
            for (int k = 0; k < NVX; ++k) { bst2(rs.p, rs.voff, soff(rs, X.reg_p, 0, k), X.p[k]); bst2(rs.p, rs.voff, soff(rs, X.reg_p, 1, k), X.r[k]); }
            X.dirty_pr = false;
        }
    }
    // Sweep 2 of the lean leapfrog (see leaf_lean): gradient at q', second half-kick, energies, level-0 criterion.
    // v = K, logp, level-0 criterion (end, start) — chain-wide sums.
    struct ParChunk { double2 mu, a, b01; double b2; };
    __device__ __forceinline__ ParChunk ldpar(const LeanRs& rs, int k) const {
        ParChunk c_;
        c_.mu = par2(rs, rs.mu, k); c_.a = par2(rs, rs.a, k); c_.b01 = par2(rs, rs.b, k); c_.b2 = par_b2(rs, k);
        return c_;
    }
    __device__ __forceinline__ void lean_sweep2(const LeanRs& rs, RegsT& X, const double h, const bool first_back, double (&v)[4]) {
        double2 accK = {0.0, 0.0}, accL = {0.0, 0.0}, accE = {0.0, 0.0}, accS = {0.0, 0.0};
        // (an opaque copy of h: with the same SSA value the compiler keeps sweep 1's half-kicked momenta of ALL chunks alive
        //  across the barrier instead of recomputing them — 4 more registers per chunk than the budget has)
        double h2 = h;
        asm volatile("" : "+v"(h2));
        // model vectors come from L2: one chunk of look-ahead (measured, profiles/r2_lean_ab_parameter_prefetch_depth.txt: three
        // chunks cost 5 % at D = 6000 and D = 10 000 — the extra live registers turn into AGPR moves)
        constexpr int PFP = 1;
        ParChunk pn[PFP];
#pragma unroll
        for (int u = 0; u < PFP; ++u) if (u < NVX) pn[u] = ldpar(rs, u);
#pragma unroll
        for (int k = 0; k < NVX; ++k) {
            const int64_t cch = (int64_t)k * W + wave;
            const ParChunk pc = pn[k % PFP];
            if (k + PFP < NVX) pn[k % PFP] = ldpar(rs, k + PFP);
            const double2 mu = pc.mu, a = pc.a, b01 = pc.b01;
            const double b2 = pc.b2;
            const double2 s2 = sigl(rs, k);
            double2 z;
            z.x = X.q[k].x - mu.x;
            z.y = X.q[k].y - mu.y;
            const double ezl = edge[2 * cch], ezr = edge[2 * cch + 3];   // (the buffer is padded with 0.0 at both ends)
            const double zl = wave_shr1(z.y, ezl), zr = wave_shl1(z.x, ezr);
            double tx = a.x * z.x;
            tx = fma(b01.x, zl, tx);
            tx = fma(b01.y, z.y, tx);
            double ty = a.y * z.y;
            ty = fma(b01.y, z.x, ty);
            ty = fma(b2, zr, ty);
            double2 gg;
            gg.x = -tx;
            gg.y = -ty;
            accL.x = fma(z.x, gg.x, accL.x);
            accL.y = fma(z.y, gg.y, accL.y);
            const double2 pold = X.p[k];
            double2 rold;
            rold.x = first_back ? -0.0 : X.r[k].x;
            rold.y = first_back ? -0.0 : X.r[k].y;
            double2 pv;
            pv.x = fma(h2, gg.x, fma(h2, X.g[k].x, pold.x));   // the first half-kick again: same operands, same bits
            pv.y = fma(h2, gg.y, fma(h2, X.g[k].y, pold.y));
            X.g[k] = gg;
            X.p[k] = pv;
            const double vx = s2.x * pv.x, vy = s2.y * pv.y;
            accK.x = fma(pv.x, vx, accK.x);
            accK.y = fma(pv.y, vy, accK.y);
            X.r[k].x = rold.x + pv.x;
            X.r[k].y = rold.y + pv.y;
            const double tx0 = (X.r[k].x - rold.x) + pold.x, ty0 = (X.r[k].y - rold.y) + pold.y;
            accE.x = fma(tx0, vx, accE.x);
            accE.y = fma(ty0, vy, accE.y);
            accS.x = fma(tx0, s2.x * pold.x, accS.x);
            accS.y = fma(ty0, s2.y * pold.y, accS.y);
            NPHIP_CHUNK_FENCE(k);
        }
        NPHIP_PHASE_FENCE();
        v[0] = accK.x + accK.y; v[1] = accL.x + accL.y; v[2] = accE.x + accE.y; v[3] = accS.x + accS.y;
        rsum(v);
        NPHIP_PHASE_FENCE();
    }

    // returns 0 (next leaf) or an end code; the caller runs the out-of-line draw end AFTER the register state is dead:
    // 1 diverged (energy error), 2 U-turn, 3 maximum depth, 4 diverged (logp not finite: no end position to report)
    __device__ __forceinline__ int leaf_lean(const LeanRs& rs, RegsT& X, Hot& H) {
        const int32_t j = H.nleaf + 1, d = H.depth, dir = H.dir;
        const int db = dir > 0 ? 1 : 0;
        const int32_t idx_new = H.idx_cur + dir;
        const int32_t srcq = H.srcq, srcp = H.srcp, newq = H.newq, newp = H.newp;
        const bool check = H.check;
#ifdef NPHIP_PROFILE
        const int64_t tp0 = (int64_t)__builtin_readcyclecounter();
#endif
        if (LRING && j == 1) X.ring_leaf1 = -1;
        // ---- source state (already resident unless the cursor moved or a rare path ran)
        if (X.reg_q != srcq) {
#pragma unroll
            for (int k = 0; k < NVX; ++k) X.q[k] = bld2(rs.q, rs.voff, soff(rs, srcq, 0, k));
            lean_grad(rs, X);
        }
        if (X.reg_p != srcp) {
#pragma unroll
            for (int k = 0; k < NVX; ++k) { X.p[k] = bld2(rs.p, rs.voff, soff(rs, srcp, 0, k)); X.r[k] = bld2(rs.p, rs.voff, soff(rs, srcp, 1, k)); }
        }
        NPHIP_PHASE_FENCE();
        // ---- leapfrog, sweep 1: q' = q + eps sigma^2 (p + eps/2 g); publish the chunk-edge z'
        const double eps = (double)dir * H.step;
        const double h = 0.5 * eps;
        const bool first_back = (idx_new == -1);   // first backward step: rho' = p'  (-0.0 + p == p exactly, also for signed zeros)
        // (software pipeline: the L2 loads of chunk k + PFM are issued before chunk k is computed; the fence at the end of every
        //  chunk keeps the scheduler from issuing them all at once — that spills the resident state)
        constexpr int PFM = 1;
        double2 mu_n[PFM];
#pragma unroll
        for (int u = 0; u < PFM; ++u) if (u < NVX) mu_n[u] = par2(rs, rs.mu, u);
#pragma unroll
        for (int k = 0; k < NVX; ++k) {
            const int64_t cch = (int64_t)k * W + wave;
            const double2 mu = mu_n[k % PFM];
            if (k + PFM < NVX) mu_n[k % PFM] = par2(rs, rs.mu, k + PFM);
            const double2 s2 = sigl(rs, k);
            const double phx = fma(h, X.g[k].x, X.p[k].x), phy = fma(h, X.g[k].y, X.p[k].y);
            X.q[k].x = fma(eps, s2.x * phx, X.q[k].x);
            X.q[k].y = fma(eps, s2.y * phy, X.q[k].y);
            const double zx = X.q[k].x - mu.x, zy = X.q[k].y - mu.y;
            if (lane == 0 || lane == 63) edge[2 * cch + (lane == 0 ? 1 : 2)] = (lane == 0) ? zx : zy;
            NPHIP_CHUNK_FENCE(k);
        }
        __syncthreads();
#ifdef NPHIP_PROFILE
        const int64_t tp_s1 = (int64_t)__builtin_readcyclecounter();
#endif
        // ---- sweep 2: gradient at q', second half-kick, energies, level-0 criterion
        // (Fusing the level-1 criteria of this leaf into the sweep — their operands behind the gradient arithmetic, one shared
        //  reduction — was built and measured against the allocator: as three instantiations or as a run-time flag it turns the
        //  resident state into a multi-way phi and spills 800+ VGPRs at 20 chunks per wave.  The criteria stay separate passes.)
        double v4[4];
        lean_sweep2(rs, X, h, first_back, v4);
        X.reg_q = newq;
        X.reg_p = newp;
        X.dirty_qg = true;
        X.dirty_pr = true;
#ifdef NPHIP_PROFILE
        const int64_t tp2 = (int64_t)__builtin_readcyclecounter();
        c->prof[0] += tp2 - tp0; c->prof[15] += tp_s1 - tp0;
#endif
        const double K = 0.5 * v4[0], lp = 0.5 * v4[1];
        const bool turn0 = (v4[2] < 0.0) || (v4[3] < 0.0);
        // ---- NutsTree::extend / merge_into, unrolled (same decisions as leaf_reg / cont_tree)
        H.nleaf += 1;
        H.n_steps += 1;
        const bool ok = isfinite(lp);
        const double Unew = -lp, E = K + Unew, dE = E - H.H0;
        const bool diverged = !ok || (dE > H.max_ee) || !isfinite(dE);
        double T_wm = 1.0;
        int64_t T_we = 0;
        if (!diverged) {
            const double x = -dE, xc = x > 1e9 ? 1e9 : (x < -1e9 ? -1e9 : x);
            double kk;
            nphip_exp_parts(xc, &T_wm, &kk);
            T_we = (int64_t)kk;
            const double e = nphip_exp_scale(x, T_wm, kk);
            const double a = e < 1.0 ? e : 1.0;
            H.acc += a;
            H.acc_sym += 2.0 * a / (1.0 + e);
        }
        if (diverged) { X.dirty_qg = X.dirty_pr = false; hot_save(H); return ok ? 1 : 4; }
#ifdef NPHIP_PROFILE
        int64_t tq = (int64_t)__builtin_readcyclecounter();
        c->prof[8] += tq - tp2;
#endif
        NPHIP_PHASE_FENCE();
        double T_U = Unew, T_E = E;
        nphip_u32x4 mrg_blk = {{0u, 0u, 0u, 0u}};
        int32_t mrg_id = -1;
        int32_t T_q = newq, T_idx = idx_new;
        H.srcq = newq; H.srcp = newp; H.idx_cur = idx_new;   // the cursor moves to the new leaf
        int32_t k = 0;
        while (k < d && (((j - 1) >> k) & 1)) {
            if (check) {
                bool turn;
                if (k == 0) {
                    turn = turn0;
                } else {
                    const int32_t a = j - (2 << k) + 1, al = j - (1 << k);
                    const int64_t sAf = first_slot_of(a, d), sAl = slot_last(__builtin_ctz((unsigned)al), A.cap);
                    if (k == 1) {
                        if (LRING && X.ring_leaf1 == al) turn = (c->pre_turn != 0) | lean_pass2<true>(rs, X, sAf, sAl);
                        else turn = (c->pre_turn != 0) | lean_pass2<false>(rs, X, sAf, sAl);
                    }
                    else turn = lean_pass3(rs, X, sAf, sAl, first_slot_of(al + 1, d));
                }
                if (turn) { X.dirty_qg = X.dirty_pr = false; hot_save(H); return 2; }
            }
#ifdef NPHIP_PROFILE
            { const int64_t t_ = (int64_t)__builtin_readcyclecounter(); c->prof[k == 0 ? 9 : 10] += t_ - tq; tq = t_; }
#endif
            NPHIP_PHASE_FENCE();
            merge_level(H, j, d, k, mrg_blk, mrg_id, T_wm, T_we, T_q, T_U, T_E, T_idx);
            NPHIP_PHASE_FENCE();
#ifdef NPHIP_PROFILE
            { const int64_t t_ = (int64_t)__builtin_readcyclecounter(); c->prof[11] += t_ - tq; tq = t_; }
#endif
            ++k;
        }
        if (k < d) {
            c->sub_wm[k] = T_wm; c->sub_we[k] = T_we; c->sub_q[k] = T_q; c->sub_U[k] = T_U; c->sub_E[k] = T_E; c->sub_idx[k] = T_idx;
            H.sub_used |= 1u << T_q;
            // (A.first, T.first) of the level-1 merge the next leaf will check: this leaf IS that T.first
#ifdef NPHIP_PROFILE
            const int64_t tp3 = (int64_t)__builtin_readcyclecounter();
#endif
            if ((j & 3) == 3 && check && d >= 2) c->pre_turn = lean_pass1(rs, X, first_slot_of(j - 2, d)) ? 1 : 0;
#ifdef NPHIP_PROFILE
            const int64_t tp4 = (int64_t)__builtin_readcyclecounter();
            c->prof[10] += tp4 - tp3;
#endif
            // ---- stores, last: q when the leaf is referenced as a candidate; (p, rho) unless the leaf is only ever the
            // source of the next leapfrog (leaf % 4 == 3)
            if (LRING && (j & 3) == 2) {
#pragma unroll
                for (int k_ = 0; k_ < NVX; ++k_) {
                    if (k_ < KP) *lring(rs, 0, k_) = X.p[k_]; else bst2(rs.p, rs.voff, soff(rs, X.reg_p, 0, k_), X.p[k_]);
                    if (k_ < KR) *lring(rs, 1, k_) = X.r[k_]; else bst2(rs.p, rs.voff, soff(rs, X.reg_p, 1, k_), X.r[k_]);
                }
                X.ring_leaf1 = j;
                lean_store(rs, X, T_q == newq, false);
            } else {
                lean_store(rs, X, T_q == newq, (j & 3) != 3);
            }
            issue_leaf_hot(H);
#ifdef NPHIP_PROFILE
            c->prof[7] += (int64_t)__builtin_readcyclecounter() - tp4;
#endif
            return 0;
        }
        // ---- the new sub-tree of depth d is complete (j == 2^d): merge into the main tree (general index modes)
        bool turn = false;
        if (check) {
            const int32_t far_slot = rfl(c->endp[1 - db]), far_idx = rfl(dir > 0 ? c->idx_left : c->idx_right);
            const int32_t near_idx = rfl(dir > 0 ? c->idx_right : c->idx_left);
            turn = lean_top(rs, X, d != 0, far_slot, rfl(c->endp[db]), slot_first((int)d), pair_of(far_idx, idx_new), pair_of(far_idx, near_idx + dir),
                            pair_of(near_idx, idx_new));
        }
        c->endq[db] = newq;
        c->endp[db] = newp;
        c->endpar[db] ^= 1;
        if (dir > 0) c->idx_right = idx_new; else c->idx_left = idx_new;
        {
            double sm; int64_t se;
            nphip_w_add(c->main_wm, c->main_we, T_wm, T_we, &sm, &se);
            const double ref = nphip_w_rel(c->main_wm, c->main_we, se), oth = nphip_w_rel(T_wm, T_we, se);
            bool take = oth >= ref;
            if (!take) take = merge_uniform_hot(H, j, d, d, mrg_blk, mrg_id) * ref < oth;
            if (take) { c->cand_q = T_q; c->cand_U = T_U; c->cand_E = T_E; c->cand_idx = T_idx; }
            c->main_wm = sm; c->main_we = se;
            H.depth = d + 1;
            c->depth = d + 1;
        }
        lean_store(rs, X, true, true);  // a new trajectory end is always written back
        if (turn) { hot_save(H); return 2; }
        if (H.depth >= A.s.maxdepth) { hot_save(H); return 3; }
        start_doubling_hot(H);
        return 0;
    }

    // memory-resident equivalent (NV == 0): the same criteria, one pass each
    __device__ __forceinline__ bool check_merge(RegsT& X, int n_checks, int64_t sA, int64_t iA, int64_t sB, int64_t iB,
                                                int64_t sTF, int64_t iTF, int64_t sTL, int64_t iTL) {
        bool turn = turning(sA, iA, sTL, iTL);
        if (n_checks == 3) {
            if (!turn) turn = turning(sB, iB, sTL, iTL);
            if (!turn) turn = turning(sA, iA, sTF, iTF);
        }
        return turn;
    }

    // EuclideanHamiltonian::is_turning on two P-slots.
    __device__ __forceinline__ bool turning(int64_t s1, int64_t i1, int64_t s2_, int64_t i2) {
        int64_t ss = s1, se = s2_, a = i1, b = i2;
        if (!(i1 < i2)) { ss = s2_; se = s1; a = i2; b = i1; }
        const double *ps = P(ss), *rs = R(ss), *pe = P(se), *re = R(se);
        const int mode = (a >= 0 && b >= 0) ? 0 : ((b >= 0 && a < 0) ? 1 : 2);
        double2 acc1 = {0.0, 0.0}, acc2 = {0.0, 0.0};
        if (BATCHED && !lr_job()) {
            struct V5 { double2 ps, rs, pe, re, s; };
            chunks(
                [&](int64_t i) { V5 v; v.ps = ld2(ps, i); v.rs = ld2(rs, i); v.pe = ld2(pe, i); v.re = ld2(re, i); v.s = sg2(i); return v; },
                [&](int64_t, const V5& v) {
                    double2 t, ve, vs;
                    ve.x = v.s.x * v.pe.x; ve.y = v.s.y * v.pe.y; vs.x = v.s.x * v.ps.x; vs.y = v.s.y * v.ps.y;
                    if (mode == 0) { t.x = (v.re.x - v.rs.x) + v.ps.x; t.y = (v.re.y - v.rs.y) + v.ps.y; }
                    else if (mode == 1) { t.x = v.re.x + v.rs.x; t.y = v.re.y + v.rs.y; }
                    else { t.x = (v.rs.x - v.re.x) + v.pe.x; t.y = (v.rs.y - v.re.y) + v.pe.y; }
                    acc1.x = fma(t.x, ve.x, acc1.x);
                    acc1.y = fma(t.y, ve.y, acc1.y);
                    acc2.x = fma(t.x, vs.x, acc2.x);
                    acc2.y = fma(t.y, vs.y, acc2.y);
                });
        } else
        NPHIP_FOR_CHUNKS(i) {
            double2 vps = ld2(ps, i), vrs = ld2(rs, i), vpe = ld2(pe, i), vre = ld2(re, i);
            double2 t, ve, vs;
            if (lr_job()) {   // the velocities M^-1 p of both ends were stored with them (the metric may be low-rank)
                ve = ld2(VEL(se), i); vs = ld2(VEL(ss), i);
            } else {
                const double2 s2 = sg2(i);
                ve.x = s2.x * vpe.x; ve.y = s2.y * vpe.y; vs.x = s2.x * vps.x; vs.y = s2.y * vps.y;
            }
            if (mode == 0) { t.x = (vre.x - vrs.x) + vps.x; t.y = (vre.y - vrs.y) + vps.y; }
            else if (mode == 1) { t.x = vre.x + vrs.x; t.y = vre.y + vrs.y; }
            else { t.x = (vrs.x - vre.x) + vpe.x; t.y = (vrs.y - vre.y) + vpe.y; }
            acc1.x = fma(t.x, ve.x, acc1.x);
            acc1.y = fma(t.y, ve.y, acc1.y);
            acc2.x = fma(t.x, vs.x, acc2.x);
            acc2.y = fma(t.y, vs.y, acc2.y);
        }
        double t1 = acc1.x + acc1.y, t2 = acc2.x + acc2.y;
        rsum2(t1, t2);
        return (t1 < 0.0) || (t2 < 0.0);
    }

    // Mass-matrix strategy init: estimators see the initial point; sig2 = 1/clamp(|g|).
    __device__ void init_mass_matrix(int64_t buf) {
        const double *q = Q(buf), *g = G(buf);
        NPHIP_FOR_CHUNKS(i) {
            double2 q2 = ld2(q, i), g2 = ld2(g, i);
            double2 s;
            if (A.s.adapt_mass_matrix) {
                double2 zero = {0.0, 0.0};
                for (int e = 0; e < 2; ++e) {
                    st2(EST(e, 0), i, q2); st2(EST(e, 1), i, zero);
                    st2(EST(e, 2), i, g2); st2(EST(e, 3), i, zero);
                }
                double vx = 1.0 / clamp_mm(fabs(g2.x)), vy = 1.0 / clamp_mm(fabs(g2.y));
                s.x = isfinite(vx) ? vx : 1.0;
                s.y = isfinite(vy) ? vy : 1.0;
            } else {
                s.x = 1.0; s.y = 1.0;
            }
            st2(sig2, i, s);
        }
        c->fg = 0;
        c->fg_count = A.s.adapt_mass_matrix ? 1 : 0;
        c->bg_count = c->fg_count;
    }

    // End-of-draw vector pass: trace row, Welford updates of both estimators, mass-matrix refresh.
    __device__ void position_pass(int64_t draw, bool do_add, bool do_switch, bool do_update, int64_t n_fg, int64_t n_bg) {
#ifdef NPHIP_PROFILE
        const int64_t tpp0_ = (int64_t)__builtin_readcyclecounter();
#endif
        const int64_t cq = c->cand_q;
        const double *q = Q(cq), *g = G(cq);
        const size_t row = ((size_t)chain * T + draw) * D;
        const int64_t efg = c->fg, ebg = 1 - c->fg;
        const int64_t esrc = do_switch ? ebg : efg;
        if (PFRARE) {
            // What the pass reads of a chunk, and what it does with it — apart, so that PFRARE has the reads of four chunks in flight ahead of their stores:
            // the plain loop waits for the stores of a chunk to COMPLETE before the next read returns (loads and stores share a counter, and a load cannot
            // pass a store to the same pool): three round trips to memory per chunk, 80 - 90 k cycles per warm-up draw at D = 1000 (round 6).
            struct PassIn { double2 q, g, s, f[4], b[4]; };
            const bool want_s = do_update || A.tr_mm != nullptr;
            auto rd = [&](int64_t i) {
                PassIn v;
                v.q = ld2(q, i); v.g = ld2(g, i);
                if (do_add) {
                    if (n_fg != 1) { v.f[0] = ld2(EST(efg, 0), i); v.f[1] = ld2(EST(efg, 1), i); v.f[2] = ld2(EST(efg, 2), i); v.f[3] = ld2(EST(efg, 3), i); }
                    if (n_bg != 1) { v.b[0] = ld2(EST(ebg, 0), i); v.b[1] = ld2(EST(ebg, 1), i); v.b[2] = ld2(EST(ebg, 2), i); v.b[3] = ld2(EST(ebg, 3), i); }
                } else if (do_update) {
                    v.f[1] = ld2(EST(esrc, 1), i);
                    v.f[3] = ld2(EST(esrc, 3), i);
                }
                if (want_s) v.s = ld2(sig2, i);
                return v;
            };
            // one estimator takes the draw in (Welford): -> its sums of squares (q, grad)
            auto welford = [&](int64_t e, int64_t n, const double2 (&in)[4], int64_t i, const double2 q2, const double2 g2, double2& vq, double2& vg) {
                double2 mq, mg;
                if (n == 1) {
                    mq = q2; mg = g2; vq.x = vq.y = 0.0; vg.x = vg.y = 0.0;
                } else {
                    const double inv = 1.0 / (double)n;
                    mq = in[0]; vq = in[1]; mg = in[2]; vg = in[3];
                    double d;
                    d = q2.x - mq.x; mq.x = fma(d, inv, mq.x); vq.x = fma(d, q2.x - mq.x, vq.x);
                    d = q2.y - mq.y; mq.y = fma(d, inv, mq.y); vq.y = fma(d, q2.y - mq.y, vq.y);
                    d = g2.x - mg.x; mg.x = fma(d, inv, mg.x); vg.x = fma(d, g2.x - mg.x, vg.x);
                    d = g2.y - mg.y; mg.y = fma(d, inv, mg.y); vg.y = fma(d, g2.y - mg.y, vg.y);
                }
                st2(EST(e, 0), i, mq); st2(EST(e, 1), i, vq); st2(EST(e, 2), i, mg); st2(EST(e, 3), i, vg);
            };
            auto body = [&](int64_t i, const PassIn& v) {
                const double2 q2 = v.q, g2 = v.g;
                if (A.tr_draws) st2_dense(A.tr_draws + row, i, D, q2);
                if (A.tr_grad) st2_dense(A.tr_grad + row, i, D, g2);
                double2 src_m2q = {0.0, 0.0}, src_m2g = {0.0, 0.0};
                if (do_add) {
                    double2 vq, vg;
                    welford(efg, n_fg, v.f, i, q2, g2, vq, vg);   // (count after adding)
                    if (efg == esrc) { src_m2q = vq; src_m2g = vg; }
                    welford(ebg, n_bg, v.b, i, q2, g2, vq, vg);
                    if (ebg == esrc) { src_m2q = vq; src_m2g = vg; }
                } else if (do_update) {
                    src_m2q = v.f[1];
                    src_m2g = v.f[3];
                }
                if (!want_s) return;
                double2 s = v.s;
                if (do_update) {
                    const int64_t n_src = do_switch ? n_bg : n_fg;
                    double vx, vy;
                    if (A.s.use_grad_based) {
                        vx = sqrt(src_m2q.x / src_m2g.x);
                        vy = sqrt(src_m2q.y / src_m2g.y);
                    } else {
                        const double scale = 1.0 / (double)(n_src - 1);
                        vx = src_m2q.x * scale;
                        vy = src_m2q.y * scale;
                    }
                    if (isfinite(vx)) s.x = clamp_mm(vx);
                    if (isfinite(vy)) s.y = clamp_mm(vy);
                    st2(sig2, i, s);
                }
                if (A.tr_mm) st2_dense(A.tr_mm + row, i, D, s);
            };
            chunks_pf<4>(rd, body);
        } else
        NPHIP_FOR_CHUNKS(i) {
            double2 q2 = ld2(q, i), g2 = ld2(g, i);
            if (A.tr_draws) st2_dense(A.tr_draws + row, i, D, q2);
            if (A.tr_grad) st2_dense(A.tr_grad + row, i, D, g2);
            double2 src_m2q = {0.0, 0.0}, src_m2g = {0.0, 0.0};
            if (do_add) {
                for (int t = 0; t < 2; ++t) {
                    const int64_t e = t == 0 ? efg : ebg;
                    const int64_t n = t == 0 ? n_fg : n_bg;  // count after adding
                    double2 mq, vq, mg, vg;
                    if (n == 1) {
                        mq = q2; mg = g2; vq.x = vq.y = 0.0; vg.x = vg.y = 0.0;
                    } else {
                        const double inv = 1.0 / (double)n;
                        mq = ld2(EST(e, 0), i); vq = ld2(EST(e, 1), i); mg = ld2(EST(e, 2), i); vg = ld2(EST(e, 3), i);
                        double d;
                        d = q2.x - mq.x; mq.x = fma(d, inv, mq.x); vq.x = fma(d, q2.x - mq.x, vq.x);
                        d = q2.y - mq.y; mq.y = fma(d, inv, mq.y); vq.y = fma(d, q2.y - mq.y, vq.y);
                        d = g2.x - mg.x; mg.x = fma(d, inv, mg.x); vg.x = fma(d, g2.x - mg.x, vg.x);
                        d = g2.y - mg.y; mg.y = fma(d, inv, mg.y); vg.y = fma(d, g2.y - mg.y, vg.y);
                    }
                    st2(EST(e, 0), i, mq); st2(EST(e, 1), i, vq); st2(EST(e, 2), i, mg); st2(EST(e, 3), i, vg);
                    if (e == esrc) { src_m2q = vq; src_m2g = vg; }
                }
            } else if (do_update) {
                src_m2q = ld2(EST(esrc, 1), i);
                src_m2g = ld2(EST(esrc, 3), i);
            }
            double2 s = ld2(sig2, i);
            if (do_update) {
                const int64_t n_src = do_switch ? n_bg : n_fg;
                double vx, vy;
                if (A.s.use_grad_based) {
                    vx = sqrt(src_m2q.x / src_m2g.x);
                    vy = sqrt(src_m2q.y / src_m2g.y);
                } else {
                    const double scale = 1.0 / (double)(n_src - 1);
                    vx = src_m2q.x * scale;
                    vy = src_m2q.y * scale;
                }
                if (isfinite(vx)) s.x = clamp_mm(vx);
                if (isfinite(vy)) s.y = clamp_mm(vy);
                st2(sig2, i, s);
            }
            if (A.tr_mm) st2_dense(A.tr_mm + row, i, D, s);
        }
#ifdef NPHIP_PROFILE
        if (!INK) c->prof[10] += (int64_t)__builtin_readcyclecounter() - tpp0_;
#endif
    }

    __device__ __forceinline__ void store_divergence(bool have_end) {
        if (!A.tr_div[0]) return;
        const size_t row = ((size_t)chain * T + c->draw) * D;
        const double *q = Q(c->lf_srcq), *g = G(c->lf_srcq), *p = P(c->lf_srcp), *qe = Q(c->lf_newq);
        NPHIP_FOR_CHUNKS(i) {
            st2_dense(A.tr_div[0] + row, i, D, ld2(q, i));
            if (have_end) st2_dense(A.tr_div[1] + row, i, D, ld2(qe, i));
            st2_dense(A.tr_div[2] + row, i, D, ld2(p, i));
            st2_dense(A.tr_div[3] + row, i, D, ld2(g, i));
        }
    }

    // Divergence record of the register-resident kernels.  The state a failed leapfrog started from lives only in registers, and
    // the leaf updates it in place — nothing of it is left when the energy error is known.  It is rebuilt here, in the rare path:
    // the state the current doubling started from (a trajectory end) is always in HBM, and the leaves of a doubling are a
    // deterministic function of it, so the j - 1 leapfrogs before the failed one are integrated again with the memory-resident
    // passes (same operations in the same order as the leaf: the same bits), then the first half of the failed one for the
    // position it ended at.  Costs a diverging draw at most its last doubling again and the hot path nothing.
    __device__ void replay_divergence(bool have_end) {
        const int64_t j = c->nleaf;   // the failed leaf (already counted)
        const int64_t dir = c->dir;
        const int db = dir > 0 ? 1 : 0;
        int64_t sq = c->endq[db], sp = c->endp[db];
        // scratch: two Q-pool buffers that are neither the draw about to be emitted nor the start state; the tree is dead, so
        // the P-slots of level 0 are free (the start state's slot is the origin's or a trajectory end's)
        int64_t b[2];
        int nb = 0;
        for (int64_t i = 0; i < A.nqpool && nb < 2; ++i)
            if (i != c->cand_q && i != sq) b[nb++] = i;
        const int64_t ps[2] = {slot_first(0), slot_last(0, A.cap)};
        double lp;
        int64_t code;
        eval_position(sq, lp, code);   // (the pool keeps q only: the gradient at the start state)
        for (int64_t i = 1; i < j; ++i) {
            const int t = (int)(i & 1);
            lf1(sq, sp, b[t], ps[t], dir);
            (void)lf2(lp, code, 1);
            sq = b[t]; sp = ps[t];
        }
        lf1(sq, sp, b[(int)(j & 1)], ps[(int)(j & 1)], dir);   // first half of the failed step: q_end
        store_divergence(have_end);
    }

    // ------------------------------------------------------------------ control flow
    __device__ void start_ss(int64_t ss_id) {
        c->ss_id = ss_id;
        c->step_size = A.s.initial_step;
        if (A.s.fixed_step_size) { da_init(A.s.initial_step); after_ss(); return; }
        double K0 = sample_momentum(NPHIP_RNG_SS_MOMENTUM, (uint32_t)ss_id);
        c->H0 = K0 + c->cand_U;
        lf1(c->cand_q, kSlotInit, alloc_q(false), slot_first(0), +1);
        c->phase = PH_SS_FIRST;
    }

    // search ids: 0xffffffff at chain start, 0x80000000 | draw when the host resumed the chain with a new metric before that draw
    // (both go on with the next draw), else the draw whose adaptation triggered the search (which is finished afterwards)
    __device__ void after_ss() {
        if (c->ss_id >= 0x80000000ll) begin_draw();
        else finish_draw();
    }

    __device__ __forceinline__ void cont_ss(double K, double lp, int64_t code, bool first) {
        c->total_steps += 1;
        const bool ok = (code == 0) && isfinite(lp);
        const double dE = (K - lp) - c->H0;
        const bool diverged = !ok || (dE > A.s.max_energy_error) || !isfinite(dE);
        if (diverged) {
            c->step_size = A.s.initial_step;
            da_init(A.s.initial_step);
            after_ss();
            return;
        }
        const double e = nphip_exp(-dE);
        const double accept = e < 1.0 ? e : 1.0;
        if (first) {
            c->ss_dir = accept > A.s.target_accept ? 1 : -1;
            c->ss_iter = 0;
        } else {
            if (c->ss_dir > 0) {
                if (accept <= A.s.target_accept || c->step_size > 1e5) { da_init(c->step_size); after_ss(); return; }
                c->step_size *= 2.0;
            } else {
                if (accept >= A.s.target_accept || c->step_size < 1e-10) { da_init(c->step_size); after_ss(); return; }
                c->step_size /= 2.0;
            }
            c->ss_iter += 1;
            if (c->ss_iter >= 100) {
                c->step_size = A.s.initial_step;
                da_init(A.s.initial_step);
                after_ss();
                return;
            }
        }
        lf1(c->cand_q, kSlotInit, c->lf_newq, slot_first(0), c->ss_dir);
        c->phase = PH_SS_ITER;
    }

    __device__ __forceinline__ void cont_init(double lp, int64_t code) {
        if (code < 0) { finish_chain(PH_ERROR, CE_FATAL_LOGP); return; }
        if (code != 0 || !isfinite(lp)) {
            // a chain resumed at a host-given position (k_resume) is never restarted from a random point behind the host's back
            if (c->init_attempt < 0) { finish_chain(PH_ERROR, CE_RESUME_FAILED); return; }
            c->init_attempt += 1;
            if (c->init_attempt >= A.s.num_try_init) { finish_chain(PH_ERROR, CE_INIT_FAILED); return; }
            gen_init(c->init_attempt);
            return;  // phase stays PH_INIT_EVAL
        }
        c->cand_q = 0;
        c->cand_U = -lp;
        init_mass_matrix(0);
        c->has_initial_mm = 1;
        c->last_update = 0;
        c->tuning = 1;
        da_init(A.s.initial_step);
        start_ss(0xffffffffll);
    }

    __device__ void begin_draw(bool now = false) {
        if (c->draw >= T) { finish_chain(PH_DONE, CE_NONE); return; }
        if (SLICED && !now) { c->phase = PH_DRAW_BEGIN; return; }   // (a slice of its own: rare_phase)
#ifdef NPHIP_PROFILE
        const int64_t tb0_ = (int64_t)__builtin_readcyclecounter();
#endif
        double K0 = sample_momentum(NPHIP_RNG_MOMENTUM, (uint32_t)c->draw);
#ifdef NPHIP_PROFILE
        if (!INK) c->prof[8] += (int64_t)__builtin_readcyclecounter() - tb0_;
#endif
        c->H0 = K0 + c->cand_U;
        c->depth = 0; c->main_wm = 1.0; c->main_we = 0;
        c->idx_left = 0; c->idx_right = 0;
        c->endq[0] = c->endq[1] = c->cand_q;
        c->endp[0] = c->endp[1] = kSlotInit;
        c->endpar[0] = c->endpar[1] = 0;
        c->cand_idx = 0; c->cand_E = c->H0;
        c->acc_sum = 0.0; c->acc_sym_sum = 0.0; c->n_steps = 0;
        start_doubling();
    }

    __device__ void start_doubling() {
        nphip_u32x4 r = nphip_philox(A.s.seed, (uint32_t)c->depth, gchain, (uint32_t)c->draw, NPHIP_RNG_DIRECTION);
        c->dir = (r.v[0] & 1u) ? 1 : -1;
        const int db = c->dir > 0 ? 1 : 0;
        c->nleaf = 0;
        c->curq = c->endq[db];
        c->curp = c->endp[db];
        c->idx_cur = c->dir > 0 ? c->idx_right : c->idx_left;
        issue_leaf();
    }

    __device__ void issue_leaf() {
        const int64_t j = c->nleaf + 1, d = c->depth;
        const int db = c->dir > 0 ? 1 : 0;
        int64_t newp;
        if (j == (1ll << d)) newp = slot_end(db, (int)(c->endpar[db] ^ 1));
        else if (j & 1) newp = (j == 1) ? slot_first((int)d) : slot_first(__builtin_ctzll((unsigned long long)(j - 1)));
        else newp = slot_last(__builtin_ctzll((unsigned long long)j), A.cap);
        lf1(c->curq, c->curp, alloc_q(true), newp, c->dir, true);
        c->phase = PH_TREE;
    }

    // one Philox block serves two merge levels of a leaf (words 0,1 for even levels, 2,3 for odd ones)
    __device__ __forceinline__ double merge_uniform(int64_t j, int64_t d, int64_t k, nphip_u32x4& blk, int64_t& blk_id) const {
        const int64_t id = k >> 1;
        if (blk_id != id) {
            const uint32_t c3 = (uint32_t)NPHIP_RNG_MERGE | ((uint32_t)d << 8) | ((uint32_t)id << 16);
            blk = nphip_philox(A.s.seed, (uint32_t)j, gchain, (uint32_t)c->draw, c3);
            blk_id = id;
        }
        return (k & 1) ? nphip_u01(blk.v[2], blk.v[3]) : nphip_u01(blk.v[0], blk.v[1]);
    }

    // the tree of the draw is complete.  Fused / resident kernels finish the draw on the spot (out of line); the launch-per-evaluation
    // kernels leave it to the next launch, which does nothing else for this chain
    __device__ __forceinline__ void draw_complete(bool diverging, bool maxdepth, bool store_div, bool div_has_end) {
        if (SLICED) {
            c->pend_end = (diverging ? 1 : 0) | (maxdepth ? 2 : 0) | (store_div ? 4 : 0) | (div_has_end ? 8 : 0);
            c->phase = PH_DRAW_END;
        } else {
            rare_end_draw(A, c, red, chain, diverging, maxdepth, store_div, div_has_end, NOG);
        }
    }
    // one leapfrog of the tree has been evaluated: NutsTree::extend / merge_into, unrolled
    // returns true when a rare, non-inlined path ran (the register mirror must then be dropped)
    // `cr` (callback kernels, leaf_cb): the criteria of every merge this leaf performs were evaluated up front, and the first half
    // of the next leapfrog may have been taken already
    __device__ __forceinline__ bool cont_tree(RegsT& X, SCacheT& Y, double K, double lp, int64_t code, bool have_turn0, bool turn0, const CbCrit* cr = nullptr) {
        if (code < 0) { finish_chain(PH_ERROR, CE_FATAL_LOGP); return true; }
        c->nleaf += 1;
        c->n_steps += 1;
        c->total_steps += 1;
        const int64_t j = c->nleaf, d = c->depth, dir = c->dir;
        const int db = dir > 0 ? 1 : 0;
        const int64_t idx_new = c->idx_cur + dir;
        const bool ok = (code == 0) && isfinite(lp);
        const double Unew = -lp, E = K + Unew, dE = E - c->H0;
        const bool diverged = !ok || (dE > A.s.max_energy_error) || !isfinite(dE);
        double T_wm = 1.0;
        int64_t T_we = 0;
        // AcceptanceRateCollector
        {
            if (!diverged) {
                // one exp serves the collector and the leaf's multinomial weight (its (p, k) parts)
                const double x = -dE, xc = x > 1e9 ? 1e9 : (x < -1e9 ? -1e9 : x);
                double kk;
                nphip_exp_parts(xc, &T_wm, &kk);
                T_we = (int64_t)kk;
                const double e = nphip_exp_scale(x, T_wm, kk);
                const double a = e < 1.0 ? e : 1.0;
                c->acc_sum += a;
                c->acc_sym_sum += 2.0 * a / (1.0 + e);
            }
        }
        if (diverged) { draw_complete(true, false, true, ok); return true; }

        const int64_t near_idx = dir > 0 ? c->idx_right : c->idx_left;
        const int64_t sT_last = c->lf_newp;
        double T_U = Unew, T_E = E;
        nphip_u32x4 mrg_blk = {{0u, 0u, 0u, 0u}};
        int64_t mrg_id = -1;
        int64_t T_q = c->lf_newq, T_idx = idx_new;
        c->curq = c->lf_newq; c->curp = c->lf_newp; c->idx_cur = idx_new;
        const bool check = A.s.check_turning && (d + 1 > A.s.mindepth);

        int64_t k = 0;
        while (k < d && (((j - 1) >> k) & 1)) {
            if (check) {
                const int64_t a = j - (2ll << k) + 1;  // first leaf of the waiting sub-tree A
                const int64_t sA_first = (a == 1) ? slot_first((int)d) : slot_first(__builtin_ctzll((unsigned long long)(a - 1)));
                const int64_t iA_first = near_idx + dir * a;
                bool turn;
                if (cr != nullptr) {
                    turn = ((cr->bits >> k) & 1u) != 0;
                } else if (k == 0) {
                    turn = have_turn0 ? turn0 : check_merge(X, 1, sA_first, iA_first, 0, 0, 0, 0, sT_last, idx_new);
                } else {
                    const int64_t al = j - (1ll << k);      // last leaf of A
                    const int64_t tf = j - (1ll << k) + 1;  // first leaf of T
                    if (FUSED && !lr_job())
                        turn = check3_stream(sA_first, slot_last(__builtin_ctzll((unsigned long long)al), A.cap),
                                             slot_first(__builtin_ctzll((unsigned long long)(tf - 1))), sT_last, Y);
                    else
                        turn = check_merge(X, 3, sA_first, iA_first, slot_last(__builtin_ctzll((unsigned long long)al), A.cap), near_idx + dir * al,
                                           slot_first(__builtin_ctzll((unsigned long long)(tf - 1))), near_idx + dir * tf, sT_last, idx_new);
                }
                if (turn) { draw_complete(false, false, false, false); return true; }
            }
            {   // multinomial merge: keep T's draw w.p. w_T / (w_A + w_T)
                double sm; int64_t se;
                nphip_w_add(c->sub_wm[k], c->sub_we[k], T_wm, T_we, &sm, &se);
                const bool take = merge_uniform(j, d, k, mrg_blk, mrg_id) * sm < nphip_w_rel(T_wm, T_we, se);
                if (!take) { T_q = c->sub_q[k]; T_U = c->sub_U[k]; T_E = c->sub_E[k]; T_idx = c->sub_idx[k]; }
                T_wm = sm; T_we = se;
            }
            ++k;
        }
        if (k < d) {
            c->sub_wm[k] = T_wm; c->sub_we[k] = T_we; c->sub_q[k] = T_q; c->sub_U[k] = T_U; c->sub_E[k] = T_E; c->sub_idx[k] = T_idx;
            if (cr != nullptr && cr->spec_q >= 0) {
                // the first half of the next leapfrog was taken inside the fused pass: what issue_leaf() / lf1() would record
                c->lf_srcq = c->curq; c->lf_srcp = c->curp; c->lf_newq = cr->spec_q; c->lf_newp = cr->spec_p; c->lf_sign = c->dir;
                c->eval_buf = cr->spec_q;
                c->phase = PH_TREE;
            } else {
                issue_leaf();
            }
            return false;
        }
        // the new sub-tree of depth d is complete: merge into the main tree
        bool turn = false;
        if (check) {
            const int64_t far_slot = c->endp[1 - db], far_idx = dir > 0 ? c->idx_left : c->idx_right;
            if (cr != nullptr) turn = cr->turn_top;
            else if (d == 0) turn = check_merge(X, 1, far_slot, far_idx, 0, 0, 0, 0, sT_last, idx_new);
            else turn = check_merge(X, 3, far_slot, far_idx, c->endp[db], near_idx, slot_first((int)d), near_idx + dir, sT_last, idx_new);
        }
        c->endq[db] = c->lf_newq;
        c->endp[db] = c->lf_newp;
        c->endpar[db] ^= 1;
        if (dir > 0) c->idx_right = idx_new; else c->idx_left = idx_new;
        {
            // biased progressive sampling at the top level: take T's draw w.p. min(1, w_T / w_main)
            double sm; int64_t se;
            nphip_w_add(c->main_wm, c->main_we, T_wm, T_we, &sm, &se);
            const double ref = nphip_w_rel(c->main_wm, c->main_we, se), oth = nphip_w_rel(T_wm, T_we, se);
            bool take = oth >= ref;
            if (!take) take = merge_uniform(j, d, d, mrg_blk, mrg_id) * ref < oth;
            if (take) { c->cand_q = T_q; c->cand_U = T_U; c->cand_E = T_E; c->cand_idx = T_idx; }
            c->main_wm = sm; c->main_we = se;
            c->depth = d + 1;
        }
        if (turn) { draw_complete(false, false, false, false); return true; }
        if (c->depth >= A.s.maxdepth) { draw_complete(false, true, false, false); return true; }
        start_doubling();
        return false;
    }

    // GlobalStrategy::adapt
    // The rare paths are compiled as separate functions that rebuild their own Machine from uniform
    // arguments: the hot Machine object never escapes, so its members stay in (S/V)GPRs.
    static __device__ __attribute__((noinline)) void rare_end_draw(const NPHIP_CONST Args& a, LdsCtl ctl, LdsDouble r, int64_t ch,
                                                                  bool diverging, bool maxdepth, bool store_div, bool div_has_end,
                                                                  bool regrad = false, bool replay = false) {
        if (W > 1) __syncthreads();   // the caller's last reduction may have used the scratch area this one starts with
        Machine m(a, ctl, r, ch);
#ifdef NPHIP_PROFILE
        const int64_t t0_ = (int64_t)__builtin_readcyclecounter();
#endif
        if (regrad) {  // the register kernel keeps only q of candidate draws in HBM: rebuild the gradient of the new point
            double lp_;
            int64_t code_;
            m.eval_position(ctl->cand_q, lp_, code_);
        }
#ifdef NPHIP_PROFILE
        ctl->prof[12] += (int64_t)__builtin_readcyclecounter() - t0_;
#endif
        if (store_div && replay) {
            if (a.tr_div[0]) m.replay_divergence(div_has_end);
        } else if (store_div) {
            if (regrad && a.tr_div[0]) {   // the divergence record wants the gradient at the start of the failed step
                double lp_;
                int64_t code_;
                m.eval_position(ctl->lf_srcq, lp_, code_);
            }
            m.store_divergence(div_has_end);
        }
        m.end_draw(diverging, maxdepth);
#ifdef NPHIP_PROFILE
        if (!INK) ctl->prof[7] += (int64_t)__builtin_readcyclecounter() - t0_;   // callback kernels: the whole draw end
#endif
        if (W > 1) __syncthreads();
    }
    static __device__ __attribute__((noinline)) void rare_phase_fn(const NPHIP_CONST Args& a, LdsCtl ctl, LdsDouble r, int64_t ch, int64_t ph) {
        if (W > 1) __syncthreads();   // (see rare_end_draw)
        Machine m(a, ctl, r, ch);
        m.rare_phase(ph);
        if (W > 1) __syncthreads();
    }

    __device__ __forceinline__ void end_draw(bool diverging, bool maxdepth) {
#ifdef NPHIP_PROFILE
        const int64_t tprof0_ = (int64_t)__builtin_readcyclecounter();
#endif
        c->fin_depth = c->depth;
        c->fin_flags = (diverging ? 1 : 0) | (maxdepth ? 2 : 0);
        c->fin_eerr = c->cand_E - c->H0;  // H0 is reused by a mid-adapt step-size search
        c->n_div += diverging ? 1 : 0;
        c->latest_steps = c->n_steps;
        // the collector's sums become the draw's means (a draw has at least one leapfrog)
        c->acc_sum = c->acc_sum / (double)c->n_steps;
        c->acc_sym_sum = c->acc_sym_sum / (double)c->n_steps;
        const bool good = diverging ? (c->cand_idx != 0) : true;
        const int64_t draw = c->draw;
        bool need_search = false;
        if (draw >= A.s.num_tune) {
            c->tuning = 0;
            position_pass(draw, false, false, false, 0, 0);
        } else if (draw < A.s.final_window) {
            const bool is_early = draw < A.s.early_end;
            const int64_t switch_freq = is_early ? A.s.early_mm_switch_freq : A.s.mm_switch_freq;
            const bool is_late = switch_freq + draw > A.s.final_window;
            bool did_change = false;
            if (A.s.adapt_mass_matrix && !c->host_metric) {   // (a chain whose metric the host supplies does not adapt its own)
                const bool do_add = good;
                const int64_t n_fg = c->fg_count + (do_add ? 1 : 0), n_bg = c->bg_count + (do_add ? 1 : 0);
                const bool do_switch = (n_bg >= switch_freq) && !is_late;
                const bool want = do_switch || (draw - c->last_update >= A.s.mm_update_freq);
                const int64_t n_src = do_switch ? n_bg : n_fg;
                const bool do_update = want && n_src >= 3;
                position_pass(draw, do_add, do_switch, do_update, n_fg, n_bg);
                if (do_switch) { c->fg = 1 - c->fg; c->fg_count = n_bg; c->bg_count = 0; }
                else { c->fg_count = n_fg; c->bg_count = n_bg; }
                if (do_update) { did_change = true; c->last_update = draw; }
            } else {
                position_pass(draw, false, false, false, 0, 0);
            }
            da_advance(is_late ? c->acc_sym_sum : c->acc_sum);
            if (did_change && c->has_initial_mm) { c->has_initial_mm = 0; need_search = true; }
            else update_stepsize(draw, false);
        } else {
            position_pass(draw, false, false, false, 0, 0);
            da_advance(c->acc_sym_sum);
            update_stepsize(draw, draw == A.s.num_tune - 1);
        }
#ifdef NPHIP_PROFILE
        c->prof[13] += (int64_t)__builtin_readcyclecounter() - tprof0_;
#endif
        if (need_search) { start_ss(draw); return; }
        finish_draw();
    }

    __device__ void finish_draw() {
        const int64_t draw = c->draw;
        if (leader()) {
            const size_t o = (size_t)chain * T + draw;
            A.st_depth[o] = c->fin_depth;
            A.st_nsteps[o] = c->latest_steps;
            A.st_idx[o] = c->cand_idx;
            A.st_diverging[o] = (uint8_t)(c->fin_flags & 1);
            A.st_maxdepth[o] = (uint8_t)((c->fin_flags >> 1) & 1);
            A.st_tuning[o] = (uint8_t)c->tuning;
            A.st_energy[o] = c->cand_E;
            A.st_energy_error[o] = c->fin_eerr;
            A.st_logp[o] = -c->cand_U;
            A.st_step[o] = c->step_size;
            A.st_step_bar[o] = nphip_exp(c->da_log_step_adapted);
            A.st_accept[o] = c->acc_sum;
            A.st_accept_sym[o] = c->acc_sym_sum;
        }
        c->draw = draw + 1;
        if (lr_job() && c->staged != 0) {
            // a metric the host handed in while this chain was running (nphip_sampler_stage_metric): taken here, between two draws of
            // the warm-up — same position, new step-size search (what PH_RESUME_SS does for a chain that had stopped for it)
            if (draw + 1 < A.s.num_tune) { apply_staged(); c->phase = PH_RESUME_SS; return; }
            c->staged = 0;
        }
        for (int i = 0; i < A.s.n_pause; ++i) {
            if (A.s.pause_draws[i] == draw + 1 && draw + 1 < T) {   // the host takes over between two draws (engine_types.h: PH_WAIT_HOST)
                c->phase = PH_WAIT_HOST;
                if (leader()) atomicAdd(&A.counters[2], 1ull);
                return;
            }
        }
#ifdef NPHIP_PROFILE
        const int64_t t0_ = (int64_t)__builtin_readcyclecounter();
#endif
        begin_draw();
#ifdef NPHIP_PROFILE
        c->prof[14] += (int64_t)__builtin_readcyclecounter() - t0_;
#endif
    }

    // The staged metric of this chain (Args::st_*) becomes its metric: the copy k_set_metric makes for a stopped chain, made by the
    // chain itself.  Every later read of (sigma^2, lr_std, lr_V, lr_lam) is a vector load of this workgroup behind the barrier /
    // program order of the rare path (as for the sigma^2 position_pass rewrites); the scalar cache is dropped for lr_lam.
    __device__ void apply_staged() {
        const int k = (int)c->staged - 1;
        const double* s2 = A.st_sig2 + (size_t)chain * ld;
        double* sd = A.lr_std + (size_t)chain * ld;
        const float* sv = A.st_V + (size_t)chain * kLrMax * ld;
        float* lv = A.lr_V + (size_t)chain * kLrMax * ld;
        NPHIP_FOR_CHUNKS(i) {
            const double2 v = ld2(s2, i);
            st2(sig2, i, v);
            double2 r; r.x = sqrt(v.x); r.y = sqrt(v.y);
            st2(sd, i, r);
            for (int j = 0; j < kLrMax; ++j) *(NPHIP_GLOBAL float2*)(lv + (size_t)j * ld + i) = *(const NPHIP_GLOBAL float2*)(sv + (size_t)j * ld + i);
        }
        if (wave == 0 && lane < kLrMax) A.lr_lam[(size_t)chain * kLrMax + lane] = A.st_lam[(size_t)chain * kLrMax + lane];
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        __builtin_amdgcn_s_dcache_inv();
        if (W > 1) __syncthreads();
        c->host_metric = 1;
        c->lr_k = k;
        c->staged = 0;
    }

    // rare phases: initial point, step-size search (memory-resident passes; not performance critical)
    __device__ __forceinline__ void rare_phase(int64_t ph) {
        double lp = 0.0;
        int64_t code = 0;
        if (ph == PH_START) {
            c->init_attempt = 0;
            gen_init(0);
            c->phase = PH_INIT_EVAL;
            if (FUSED) chain_sync<W>();
        } else if (ph == PH_RESUME_SS) {
            // the host replaced the metric between two draws (nphip_sampler_set_metric): same position, new step-size search
            c->has_initial_mm = 0;
            start_ss(0x80000000ll | c->draw);
        } else if (ph == PH_DRAW_BEGIN) {
            begin_draw(true);
        } else if (ph == PH_INIT_EVAL) {
            eval_position(c->eval_buf, lp, code);
            cont_init(lp, code);
            if (FUSED && c->phase == PH_INIT_EVAL) chain_sync<W>();
        } else {  // PH_SS_FIRST / PH_SS_ITER
            double K = lf2(lp, code, c->lf_sign);
            if (code < 0) { finish_chain(PH_ERROR, CE_FATAL_LOGP); return; }
            cont_ss(K, lp, code, ph == PH_SS_FIRST);
        }
    }

    __device__ __forceinline__ void run(int budget, bool have, LdsDouble sig_copy = nullptr, bool materialise = false) {
        if (!INK && materialise && c->phase == PH_TREE) {
            // the previous launch was a resident one (its tree leapfrogs are issued deferred): do the first half now, so that
            // this launch finds what a callback launch expects — an evaluation pending
            lf1(c->lf_srcq, c->lf_srcp, c->lf_newq, c->lf_newp, c->lf_sign);
        }
        bool worked = false;   // SLICED: this launch has advanced the chain already
        for (;;) {
            const int64_t ph = c->phase;
            if (ph == PH_DONE || ph == PH_ERROR || ph == PH_WAIT_HOST) {
                // a resident launch: the group's rendezvous counts every chain, so a finished one keeps answering the roll
                if (REMOTE && !DENS && !DG) { while (c->hs_last == 0) remote_sync(); }
                break;   // (DG: the remaining rounds of the launch are taken as a ghost, in k_advance)
            }
            if (SLICED && (ph == PH_DRAW_END || ph == PH_DRAW_BEGIN)) {
                // a slice of the end of a draw: a launch's worth of work by itself, and it consumes no evaluation (whatever the
                // callback made of this chain's stale row is ignored)
                if (worked) break;
                if (ph == PH_DRAW_END) {
                    const int64_t pe = c->pend_end;
                    rare_end_draw(A, c, red, chain, (pe & 1) != 0, (pe & 2) != 0, (pe & 4) != 0, (pe & 8) != 0, NOG);
                } else {
                    rare_phase_fn(A, c, red, chain, ph);
                }
                break;
            }
            if (ph != PH_START && ph != PH_RESUME_SS) {   // (those two consume no evaluation)
                if (REMOTE && !DENS && !DG) {
                    if (c->hs_last != 0) break;   // the host asked the launch to end at this boundary
                } else if (FUSED || DENS || DG) {
                    if (budget <= 0) break;
                    --budget;
                } else {
                    if (!have) break;
                    have = false;
                }
            }
            worked = true;
            if (ph != PH_TREE) {
                rare_phase_fn(A, c, red, chain, ph);
                continue;
            }
            // ---- a run of consecutive tree leaves.  The register mirror / streaming cache live exactly as long as
            // the run: a rare (non-inlined) path ends it, so nothing has to be kept alive across those calls.
            RegsT X;
            SCacheT Y;
            Hot H;        // register-resident kernels (leaf_reg): the control words of the leaf loop
            constexpr bool HOT = NV > 0;
            if (HOT) hot_load(H);
            int32_t end_code_ = 0;   // (ENDOUT: how the leaf that ended the draw ended it)
            bool rare = false, out_of_budget = false;
            int lean_end = 0;
            const LeanRs lrs = lean_rs();
            if ((NV == 0 || NV == -1 || LEAN) && sig_copy != nullptr) {
                // stage sigma^2 in LDS: it only changes in the rare draw-end path
                for (int64_t i = 2 * (int64_t)threadIdx.x; i < ld; i += 2 * (int64_t)blockDim.x)
                    *(NPHIP_LDS double2*)(sig_copy + i) = ld2(sig2, i);
                __syncthreads();
                sig_lds = sig_copy;
            }
            for (;;) {
#ifdef NPHIP_PROFILE
                const int64_t t0 = (int64_t)__builtin_readcyclecounter();
#endif
                if (NV > 0) {
                    if (LEAN) { lean_end = leaf_lean(lrs, X, H); rare = lean_end != 0; }
                    else rare = leaf_reg(X, H, end_code_);
                } else {
                    double lp = 0.0;
                    int64_t code = 0;
                    if (FUSED && !lr_job()) {
                        bool turn0 = false;
                        const bool even_leaf = ((c->nleaf + 1) & 1) == 0;
                        const double K = lf_stream(lp, c->idx_cur + c->dir, turn0, Y);
                        rare = cont_tree(X, Y, K, lp, code, even_leaf, turn0);
                    } else {
                        // (fused models under the low-rank metric: the two-pass leapfrog of the callback kernels, with the gradient
                        //  evaluated in lf2; the deferred first half is performed here)
                        if (FUSED) lf1(c->lf_srcq, c->lf_srcp, c->lf_newq, c->lf_newp, c->lf_sign);
                        CbCrit cr;
                        bool fast = false;
                        if (cb_fast_ok()) {   // callback kernels, rows of up to CBK chunks per wave: the fused leaf (an evaluation that failed: the plain passes)
                            lp = A.ueval[chain];
                            code = A.ecode ? A.ecode[chain] : 0;
                            fast = code == 0 && isfinite(lp);
                        }
                        const double K = fast ? leaf_cb(cr) : lf2(lp, code, c->idx_cur + c->dir);
#ifdef NPHIP_PROFILE
                        c->prof[0] += (int64_t)__builtin_readcyclecounter() - t0;
#endif
                        rare = cont_tree(X, Y, K, lp, code, false, false, fast ? &cr : nullptr);
                    }
                }
#ifdef NPHIP_PROFILE
                const int64_t t2 = (int64_t)__builtin_readcyclecounter();
                c->prof[3] += 1;
                if (rare) { c->prof[2] += t2 - t0; c->prof[5] += 1; } else { c->prof[1] += t2 - t0; c->prof[4] += 1; }
#endif
                if (rare || (!HOT && c->phase != PH_TREE)) break;   // (leaf_reg leaves the tree only through a rare path)
                // the next leaf of the run: same admission test as at the top of the outer loop
                if (REMOTE && !DENS && !DG) {
                    if (c->hs_last != 0) { out_of_budget = true; break; }
                } else if (FUSED || DENS || DG) {
                    if (budget <= 0) { out_of_budget = true; break; }
                    --budget;
                } else {
                    out_of_budget = true;   // callbacks: one evaluation per launch
                    break;
                }
            }
            if (!rare) {   // launch boundary: registers (and the LDS slot) that hold the only copy of tree state go back to memory
                if (HOT) hot_save(H);
                flush(X);
            }
            sig_lds = nullptr;
            if (ENDOUT && end_code_ != 0) {
                const int32_t ec = end_code_;
                rare_end_draw(A, c, red, chain, (ec & 2) != 0, (ec & 4) != 0, (ec & 8) != 0, (ec & 16) != 0, (ec & 32) != 0, (ec & 64) != 0);
            }
            if (LEAN && lean_end != 0) rare_end_draw(A, c, red, chain, lean_end == 1 || lean_end == 4, lean_end == 3, lean_end == 1 || lean_end == 4, lean_end == 1, true, true);
            if (out_of_budget) break;
        }
    }
#undef NPHIP_FOR_CHUNKS
};

// `Ap` points to the engine's argument block in device memory (written once at set-up); it is read through
// the constant address space, i.e. with scalar loads into SGPRs.  Per-launch scalars are kernel parameters.
// WIDE: the same kernel without the two-waves-per-SIMD register cap of the small one-wave kernels — what a job that brings at most one
// wave per SIMD (<= 1024 chains) is launched with: 2 / 3 chunks per lane spill 24 / 75 VGPRs under the cap (D = 256 / 384, 1024 chains:
// 385 -> 424 / 297 -> 378 M leapfrogs/s without it); with 4096 chains the second wave per SIMD is worth more (616 against 437 / 404
// against 389: profiles/r5_small_kernels_register_cap.txt)
// DENSEG: the resident form of the dense-precision Gaussian (REMOTE with the launch-wide GEMM as the evaluation: Machine<..., DG>, dg_round)
template <bool FUSED, int W, int NV, bool LEAN = false, bool REMOTE = false, bool LR = false, bool WIDE = false, bool DENSEG = false>
__global__ __launch_bounds__(W == 1 ? 256 : 64 * W) __attribute__((amdgpu_waves_per_eu(LEAN ? NPHIP_LEAN_OCC(W) : ((NV > 0 && !LR && !WIDE && NV <= (W == 1 ? NPHIP_W1_OCC2_MAX : NPHIP_RW_OCC2_MAX)) ? 2 : ((!FUSED && NV == 0 && !REMOTE) ? NPHIP_CB_OCC(W) : 1)), LEAN ? NPHIP_LEAN_OCC(W) : ((NV > 0 && !LR && !WIDE && NV <= (W == 1 ? NPHIP_W1_OCC2_MAX : NPHIP_RW_OCC2_MAX)) ? 2 : 8)))) void k_advance(const Args* __restrict__ Ap, int max_evals, int have_result, const LaunchSlice sl) {
    const NPHIP_CONST Args& A = *(const NPHIP_CONST Args*)Ap;
    constexpr int WAVES = (W == 1) ? 4 : W;
    __shared__ Ctl s_ctl[WAVES];
    __shared__ double s_red[W == 1 ? 8 : 32 * WAVES];   // two alternating reduction areas of 16 values per wave (one wave per chain: DPP only)
    __shared__ __attribute__((aligned(16))) double s_par[(NV > 0 && W == 1) ? 3 * 128 * NV + 8 : 2];
    __shared__ __attribute__((aligned(16))) double s_ring[(NV > 0 && !LEAN && !LR && !(W == 1 && NV > 8)) ? WAVES * 4 * 128 * NV : 2];  // per wave: 2 slots x (p, rho)
    __shared__ double s_edge[(NV > 0 && W > 1) ? 2 * W * NV + 2 : 2];   // lean kernels: padded with one 0.0 at each end
    extern __shared__ __attribute__((aligned(16))) double s_dyn[];  // a.sig_lds: sigma^2 of the chain [ld]
    const int wib = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    // a launch covers the chains [sl.chain_lo, sl.chain_lo + sl.chain_n): all of them, or one group of a pipelined host-callback job
    const int cpb = (W == 1 && REMOTE && NPHIP_JIT != 0) ? sl.cpb : 4;   // chains of this workgroup (LaunchSlice::cpb)
    const int64_t chain = sl.chain_lo + ((W == 1) ? (int64_t)blockIdx.x * cpb + wib : (int64_t)blockIdx.x);
    if (NV > 0 && W == 1 && !REMOTE) {
        // stage the fused model in LDS once per workgroup: mu | a | b shifted by one with -0.0 sentinels
        const int64_t ld = A.ld;
        NPHIP_LDS double* sp = (NPHIP_LDS double*)s_par;
        for (int64_t i = threadIdx.x; i < ld; i += blockDim.x) {
            sp[i] = ld1(A.m_mu, i);
            sp[ld + i] = ld1(A.m_a, i);
            sp[2 * ld + i] = (i == 0) ? -0.0 : ld1(A.m_b, i - 1);
        }
        if (threadIdx.x < 8) sp[3 * ld + threadIdx.x] = -0.0;
        __syncthreads();
    }
#if NPHIP_JIT
    if (REMOTE) {   // the model's shared LDS block: filled once per workgroup and launch, by all of its threads
        if (threadIdx.x == 0) nphip_chains_per_block_ = cpb;   // (NPHIP_CHAIN_SLOT of the generated prelude)
        nphip_density_stage(*(const NphipData*)A.dens_data, (double*)s_dyn + (size_t)(W == 1 ? 4 : 1) * A.dens_lds_doubles, (int)threadIdx.x, (int)blockDim.x);
        __syncthreads();
    }
#endif
    if (DENSEG) dg_register(A);
    // (DENSEG: a wave without a chain — the last workgroup's spare waves — still takes part in every round of the launch-wide GEMM)
    const bool dg_spare = DENSEG && chain >= (int64_t)sl.chain_lo + sl.chain_n;
    if (!DENSEG && (chain >= (int64_t)sl.chain_lo + sl.chain_n || (W == 1 && wib >= cpb))) return;
    if (REMOTE && !NPHIP_JIT) {
        // Roll call: chains of a resident launch wait for each other inside the kernel, so all of them must be on the device
        // before any starts.  Each arrives once; the last one sets the verdict GO.  A chain that has waited 5 ms sets it to FAIL
        // (the device is busy with something that does not end: e.g. another sampler's resident launch) and everybody — also
        // the chains that only get scheduled later — leaves without having touched anything; the host then falls back to one
        // launch per evaluation.
        unsigned long long* verdict = A.grp_go_dev + 1;
        unsigned* cnt = (unsigned*)(A.grp_go_dev + 2);
        const unsigned long long mine = (unsigned long long)sl.seq << 8;
        unsigned state = 0;
        if (lane == 0 && (W == 1 || wib == 0)) {   // (one arrival per chain)
            unsigned long long v = __hip_atomic_load(verdict, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if ((v >> 8) == (unsigned long long)sl.seq) {
                state = (unsigned)(v & 0xff);
            } else if (!dg_spare && atomicAdd(cnt, 1u) + 1u == (unsigned)sl.chain_n) {
                atomicExch(cnt, 0u);
                for (;;) {
                    if (atomicCAS(verdict, v, mine | kRollGo) == v) { state = (unsigned)kRollGo; break; }
                    v = __hip_atomic_load(verdict, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if ((v >> 8) == (unsigned long long)sl.seq) { state = (unsigned)(v & 0xff); break; }
                }
            } else {
                const long long t0 = wall_clock64();   // 100 MHz
                for (;;) {
                    v = __hip_atomic_load(verdict, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if ((v >> 8) == (unsigned long long)sl.seq) { state = (unsigned)(v & 0xff); break; }
                    if (wall_clock64() - t0 > 500000ll && atomicCAS(verdict, v, mine | kRollFail) == v) {
                        A.grp_flag[3] = (unsigned long long)sl.seq;
                        __threadfence_system();
                        state = (unsigned)kRollFail;
                        break;
                    }
                    __builtin_amdgcn_s_sleep(16);
                }
            }
        }
        state = (unsigned)__builtin_amdgcn_readfirstlane((int)state);
        if (W > 1) {   // the chain's other waves take wave 0's verdict
            NPHIP_LDS unsigned* sh = (NPHIP_LDS unsigned*)s_red;
            if (threadIdx.x == 0) sh[0] = state;
            __syncthreads();
            state = sh[0];
            __syncthreads();
        }
        if (state != (unsigned)kRollGo) return;
    }
    if (DENSEG) dg_enroll(A);
    if (DENSEG && dg_spare) {
        for (int r = 0; r < max_evals; ++r) dg_round(A, r, (int64_t)sl.chain_n);
        return;
    }
    if (LEAN && threadIdx.x == 0) { s_edge[0] = 0.0; s_edge[2 * W * NV + 1] = 0.0; }   // (the first barrier is in the sigma^2 staging)
    LdsCtl c = (LdsCtl)&s_ctl[wib];
#ifdef NPHIP_PROFILE
    const int64_t tk0_ = (int64_t)__builtin_readcyclecounter();
#endif
    {
        const NPHIP_GLOBAL uint64_t* src = (const NPHIP_GLOBAL uint64_t*)(A.ctl + chain);
        NPHIP_LDS uint64_t* dst = (NPHIP_LDS uint64_t*)c;
        for (int w = lane; w < kCtlWords; w += 64) dst[w] = src[w];
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    if (REMOTE && !NPHIP_JIT) {
        int g = 0;
        for (int o = 1; o < sl.n_grp; ++o) g += ((int)(chain - sl.chain_lo) >= sl.grp_lo[o]) ? 1 : 0;
        c->hs_seq = (int64_t)sl.grp_seq[g]; c->hs_last = 0; c->hs_grp = g; c->hs_n = sl.grp_lo[g + 1] - sl.grp_lo[g];
        const int64_t left = (int64_t)sl.chain_n - (int64_t)blockIdx.x * 4;
        c->hs_wgn = (W > 1) ? 1 : (left < 4 ? left : 4);   // chains of this workgroup (W == 1: four; group bounds are multiples of 4)
        if (DENSEG) { c->hs_seq = 0; c->hs_n = max_evals; }    // rounds of the launch-wide GEMM taken / to take
    }
    Machine<FUSED, W, NV, LEAN, REMOTE, LR, (DENSEG ? 2 : (WIDE ? 1 : 0))> m(A, c, (LdsDouble)s_red, chain, (LdsDouble)s_par, (LdsDouble)s_ring + ((LEAN || LR || (W == 1 && NV > 8)) ? 0 : (size_t)wib * 4 * 128 * (NV > 0 ? NV : 1)),
                            (LdsDouble)s_edge);
    __shared__ double s_park[(!FUSED && NV == 0 && W > 1) ? Machine<FUSED, W, NV, LEAN, REMOTE, LR, (DENSEG ? 2 : (WIDE ? 1 : 0))>::kParkMax * W : 2];
    m.parked = (LdsDouble)s_park;
    m.run(max_evals, have_result != 0, (LEAN || ((NV == 0 || NV == -1) && W >= 8 && A.sig_lds)) ? (LdsDouble)s_dyn : nullptr, sl.materialise != 0);
    if (DENSEG) {   // the rounds this chain did not need (it finished, stopped, or its last steps consumed no evaluation): as a ghost
        while (c->hs_seq < (int64_t)max_evals) { dg_round(A, (int)c->hs_seq, (int64_t)sl.chain_n, (NPHIP_LDS int64_t*)c->prof); c->hs_seq = c->hs_seq + 1; }
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
#ifdef NPHIP_PROFILE
    if (!FUSED && NV == 0) c->prof[15] += (int64_t)__builtin_readcyclecounter() - tk0_;   // callback kernels: the chain's whole launch
#endif
    if (W == 1 || wib == 0) {
        NPHIP_GLOBAL uint64_t* dst = (NPHIP_GLOBAL uint64_t*)(A.ctl + chain);
        const NPHIP_LDS uint64_t* src = (const NPHIP_LDS uint64_t*)c;
        for (int w = lane; w < kCtlWords; w += 64) dst[w] = src[w];
    }
    if (sl.grp >= 0 && !REMOTE) {
        // Pipelined host-callback groups: the host does not synchronise with the stream, it polls a word in pinned host memory.
        // Every chain of the group arrives once (its positions / control block are written and released to the system first);
        // the last arriver publishes the job-wide done / error counts and then the sequence number of this launch.
        if (W > 1) __syncthreads();
        if (lane == 0 && (W == 1 || wib == 0)) {
            __threadfence_system();
            const unsigned old = atomicAdd(&A.grp_arrive[sl.grp], 1u);
            if (old + 1u == (unsigned)sl.chain_n) {
                A.grp_arrive[sl.grp] = 0u;
                volatile unsigned long long* f = A.grp_flag + 4 * sl.grp;
                f[1] = atomicAdd(&A.counters[0], 0ull);
                f[2] = atomicAdd(&A.counters[1], 0ull);
                __threadfence_system();
                f[0] = (unsigned long long)sl.seq;
                __threadfence_system();
            }
        }
    }
}

// ----------------------------------------------------------------------------------------
// launch tables.  The file is compiled as twelve translation units in parallel (Makefile: -DNPHIP_PART=0..6, 8..12), each instantiating
// one family of kernels; without NPHIP_PART (developer builds, see the NPHIP_DEV_* macros) everything is in one.
// ----------------------------------------------------------------------------------------
#ifndef NPHIP_PART
#define NPHIP_PART -1
#endif
#define NPHIP_HAS(p) (NPHIP_PART == -1 || NPHIP_PART == (p))
#if defined(NPHIP_DEV_LEAN) || defined(NPHIP_DEV_W1NV) || defined(NPHIP_DEV_CB_W) || defined(NPHIP_DEV_W1NV_LR) || defined(NPHIP_DEV_RW_LR_W) || defined(NPHIP_DEV_DG_NV)
#define NPHIP_DEV_BUILD 1   // one kernel instantiation only: seconds instead of minutes
#endif

hipError_t launch_fam_w1(const Args& a, const Args* d_args, hipStream_t st, const LaunchSlice sl);              // part 0
hipError_t launch_fam_w1_lr(const Args& a, const Args* d_args, hipStream_t st, const LaunchSlice sl);           // part 8
hipError_t launch_fam_lean4(const Args& a, const Args* d_args, hipStream_t st, const LaunchSlice sl);           // part 1
hipError_t launch_fam_lean8(const Args& a, const Args* d_args, hipStream_t st, const LaunchSlice sl);           // part 2
hipError_t launch_fam_rw(const Args& a, const Args* d_args, int W, hipStream_t st, const LaunchSlice sl);       // part 3
hipError_t launch_fam_rw_lr(const Args& a, const Args* d_args, int W, hipStream_t st, const LaunchSlice sl);    // part 9
hipError_t launch_fam_mem(const Args& a, const Args* d_args, bool fused, int W, hipStream_t st, const LaunchSlice sl);   // part 4

// dynamic LDS of the lean kernels: sigma^2, and (4 waves per chain) as much of one (p, rho) summary as fits beside it (Machine::LR_FREE)
static size_t lean_dyn_lds(const Args& a, int W) {
    size_t dyn = (size_t)a.ld * 8;
    if (W == 4) {
        const long chunk_bytes = 4 * 1024, free_chunks = (163840 - (a.reg_nv > 20 ? 12288 : 8192) - (long)a.reg_nv * chunk_bytes) / chunk_bytes;   // (= Machine::LR_FREE)
        dyn += (size_t)std::max(0l, std::min(free_chunks, 2l * a.reg_nv)) * chunk_bytes;
    }
    return dyn;
}
#define NPHIP_LAUNCH_LEAN(WW, NN) hipLaunchKernelGGL((k_advance<true, WW, NN, true>), g, b, dyn, st, d_args, a.max_evals, a.have_result, sl)

#if NPHIP_HAS(1)
// lean register-resident kernels, 4 waves per chain (4096 < D <= 12288): state spread over VGPRs + AGPRs (one wave per SIMD), 9..24
// chunks per wave; one workgroup = one chain
hipError_t launch_fam_lean4(const Args& a, const Args* d_args, hipStream_t st, const LaunchSlice sl) {
    const dim3 g((unsigned)sl.chain_n), b(256);
    const size_t dyn = lean_dyn_lds(a, 4);
    (void)g; (void)b; (void)dyn;
    switch (a.reg_nv) {
#if defined(NPHIP_DEV_LEAN) && defined(NPHIP_DEV_W) && defined(NPHIP_DEV_NC)
#if NPHIP_DEV_W == 4
        case NPHIP_DEV_NC: NPHIP_LAUNCH_LEAN(4, NPHIP_DEV_NC); break;
#endif
#elif !defined(NPHIP_DEV_BUILD)
        case 9: NPHIP_LAUNCH_LEAN(4, 9); break;
        case 10: NPHIP_LAUNCH_LEAN(4, 10); break;
        case 11: NPHIP_LAUNCH_LEAN(4, 11); break;
        case 12: NPHIP_LAUNCH_LEAN(4, 12); break;
        case 13: NPHIP_LAUNCH_LEAN(4, 13); break;
        case 14: NPHIP_LAUNCH_LEAN(4, 14); break;
        case 15: NPHIP_LAUNCH_LEAN(4, 15); break;
        case 16: NPHIP_LAUNCH_LEAN(4, 16); break;
        case 17: NPHIP_LAUNCH_LEAN(4, 17); break;
        case 18: NPHIP_LAUNCH_LEAN(4, 18); break;
        case 19: NPHIP_LAUNCH_LEAN(4, 19); break;
        case 20: NPHIP_LAUNCH_LEAN(4, 20); break;
        // 21 .. 24 chunks per wave (10 240 < D <= 12 288, round 6): the state no longer fits the 512 registers of a lane — 16 per chunk beside
        // a working set of ~130 — and the build spills (22 chunks: 216 bytes of scratch per lane, 24: 456); still 1.6 x / 1.3 x the
        // memory-resident kernels that ran these rows before (D = 11 264: 8.6 against 5.4 M leapfrogs/s, D = 12 000: 6.5 against 5.0;
        // profiles/r6_lean4_beyond_20_chunks.txt)
        case 21: NPHIP_LAUNCH_LEAN(4, 21); break;
        case 22: NPHIP_LAUNCH_LEAN(4, 22); break;
        case 23: NPHIP_LAUNCH_LEAN(4, 23); break;
        case 24: NPHIP_LAUNCH_LEAN(4, 24); break;
#endif
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}
#endif

#if NPHIP_HAS(2)
// lean register-resident kernels, 8 waves per chain (on request: waves_per_chain = 8), 1..10 chunks per wave
hipError_t launch_fam_lean8(const Args& a, const Args* d_args, hipStream_t st, const LaunchSlice sl) {
    const dim3 g((unsigned)sl.chain_n), b(512);
    const size_t dyn = lean_dyn_lds(a, 8);
    (void)g; (void)b; (void)dyn;
    switch (a.reg_nv) {
#if defined(NPHIP_DEV_LEAN) && defined(NPHIP_DEV_NC)
#if !defined(NPHIP_DEV_W) || NPHIP_DEV_W == 8
        case NPHIP_DEV_NC: NPHIP_LAUNCH_LEAN(8, NPHIP_DEV_NC); break;
#endif
#elif !defined(NPHIP_DEV_BUILD)
        case 1: NPHIP_LAUNCH_LEAN(8, 1); break;
        case 2: NPHIP_LAUNCH_LEAN(8, 2); break;
        case 3: NPHIP_LAUNCH_LEAN(8, 3); break;
        case 4: NPHIP_LAUNCH_LEAN(8, 4); break;
        case 5: NPHIP_LAUNCH_LEAN(8, 5); break;
        case 6: NPHIP_LAUNCH_LEAN(8, 6); break;
        case 7: NPHIP_LAUNCH_LEAN(8, 7); break;
        case 8: NPHIP_LAUNCH_LEAN(8, 8); break;
        case 9: NPHIP_LAUNCH_LEAN(8, 9); break;
        case 10: NPHIP_LAUNCH_LEAN(8, 10); break;
#endif
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}
#endif
#undef NPHIP_LAUNCH_LEAN

#if NPHIP_HAS(3)
// register-resident, several waves per chain (1024 < D <= 4096, or fewer chains than SIMDs): one workgroup = one chain
hipError_t launch_fam_rw(const Args& a, const Args* d_args, int W, hipStream_t st, const LaunchSlice sl) {
#define NPHIP_LAUNCH_RW(WW, NN) hipLaunchKernelGGL((k_advance<true, WW, NN>), g, b, 0, st, d_args, a.max_evals, a.have_result, sl)
#if defined(NPHIP_DEV_RW_W) && defined(NPHIP_DEV_RW_NV)
    const dim3 g((unsigned)sl.chain_n), b(64 * W);
    if (W != NPHIP_DEV_RW_W || a.reg_nv != NPHIP_DEV_RW_NV) return hipErrorInvalidValue;
    NPHIP_LAUNCH_RW(NPHIP_DEV_RW_W, NPHIP_DEV_RW_NV);
    return hipGetLastError();
#elif defined(NPHIP_DEV_BUILD)
    return hipErrorInvalidValue;
#else
    const dim3 g((unsigned)sl.chain_n), b(64 * W);
    if (W == 2) switch (a.reg_nv) {
        case 1: NPHIP_LAUNCH_RW(2, 1); break;
        case 2: NPHIP_LAUNCH_RW(2, 2); break;
        case 3: NPHIP_LAUNCH_RW(2, 3); break;
        case 4: NPHIP_LAUNCH_RW(2, 4); break;
        case 5: NPHIP_LAUNCH_RW(2, 5); break;
        case 6: NPHIP_LAUNCH_RW(2, 6); break;
        case 7: NPHIP_LAUNCH_RW(2, 7); break;
        case 8: NPHIP_LAUNCH_RW(2, 8); break;
        default: return hipErrorInvalidValue;
    } else switch (a.reg_nv) {
        case 1: NPHIP_LAUNCH_RW(4, 1); break;
        case 2: NPHIP_LAUNCH_RW(4, 2); break;
        case 3: NPHIP_LAUNCH_RW(4, 3); break;
        case 4: NPHIP_LAUNCH_RW(4, 4); break;
        case 5: NPHIP_LAUNCH_RW(4, 5); break;
        case 6: NPHIP_LAUNCH_RW(4, 6); break;
        case 7: NPHIP_LAUNCH_RW(4, 7); break;
        case 8: NPHIP_LAUNCH_RW(4, 8); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
#endif
#undef NPHIP_LAUNCH_RW
}
#endif

// The one-wave kernels with 2 .. 8 chunks per lane (128 < D <= 1024: the ENDOUT family, the headline's among them) are a translation unit of their
// own (part 12), compiled WITHOUT inter-procedural register allocation (Makefile: -mllvm -enable-ipra=0): their leaf makes no call, and under the
// plain calling convention what lives across the kernel's few calls sits in callee-saved registers whatever the callees use.  Same-box A/B per
// kernel (warm-up, 1024 chains, M leapfrogs/s): 8 chunks 223.1 224.5 -> 228.5 230.0 (4 / 18 spilled VGPRs / SGPRs instead of 0 / 37), 6: 264.9 260.4 ->
// 266.5 265.3, 4: 345.6 343.1 -> 345.6 346.6, 3: 377.2 374.3 -> 381.1 383.8, 2: 417.7 418.6 -> 418.2 420.4; bit-identical.  Also the dense Gaussian's
// resident kernels (part 11: 13.89 -> 14.00 M leapfrogs/s) and the compiled densities (density.py: config 3 60.7 -> 61.0, two / four waves per chain
// +2.4 % / +5 %).  NOT the lean kernels (D = 10 000: 15.4 -> 12.6) and not the low-rank leaf (D = 1000, k = 16: 16.1 -> 14.7).
// (profiles/r6_call_placement_and_draw_end.txt D)
hipError_t launch_w1_noipra(int nv, const Args* d_args, hipStream_t st, const LaunchSlice sl, int me, int hr);
#if NPHIP_HAS(12) && !defined(NPHIP_DEV_BUILD)
hipError_t launch_w1_noipra(int nv, const Args* d_args, hipStream_t st, const LaunchSlice sl, int me, int hr) {
    const dim3 g(((unsigned)sl.chain_n + 3) / 4), b(256);
    switch (nv) {
        // (2, 3 chunks per lane: built for two waves per SIMD — unless the job has no second wave to bring: k_advance<..., WIDE>)
        case 2: if (sl.chain_n <= 1024) hipLaunchKernelGGL((k_advance<true, 1, 2, false, false, false, true>), g, b, 0, st, d_args, me, hr, sl);
                else hipLaunchKernelGGL((k_advance<true, 1, 2>), g, b, 0, st, d_args, me, hr, sl);
                break;
        case 3: if (sl.chain_n <= 1024) hipLaunchKernelGGL((k_advance<true, 1, 3, false, false, false, true>), g, b, 0, st, d_args, me, hr, sl);
                else hipLaunchKernelGGL((k_advance<true, 1, 3>), g, b, 0, st, d_args, me, hr, sl);
                break;
        case 4: hipLaunchKernelGGL((k_advance<true, 1, 4>), g, b, 0, st, d_args, me, hr, sl); break;
        case 5: hipLaunchKernelGGL((k_advance<true, 1, 5>), g, b, 0, st, d_args, me, hr, sl); break;
        case 6: hipLaunchKernelGGL((k_advance<true, 1, 6>), g, b, 0, st, d_args, me, hr, sl); break;
        case 7: hipLaunchKernelGGL((k_advance<true, 1, 7>), g, b, 0, st, d_args, me, hr, sl); break;
        case 8: hipLaunchKernelGGL((k_advance<true, 1, 8>), g, b, 0, st, d_args, me, hr, sl); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}
#endif

#if NPHIP_HAS(0)
// register-resident, one wave per chain (D <= 1024): four chains per workgroup, one instantiation per exact chunk count
hipError_t launch_fam_w1(const Args& a, const Args* d_args, hipStream_t st, const LaunchSlice sl) {
    const dim3 g(((unsigned)sl.chain_n + 3) / 4), b(256);
    const int me = a.max_evals, hr = a.have_result;
    (void)g; (void)b; (void)me; (void)hr;
    switch (a.reg_nv) {
#if defined(NPHIP_DEV_W1NV)
        case NPHIP_DEV_W1NV: hipLaunchKernelGGL((k_advance<true, 1, NPHIP_DEV_W1NV>), g, b, 0, st, d_args, me, hr, sl); break;
#elif !defined(NPHIP_DEV_BUILD)
        case 1: hipLaunchKernelGGL((k_advance<true, 1, 1>), g, b, 0, st, d_args, me, hr, sl); break;
        case 2: case 3: case 4: case 5: case 6: case 7: case 8: return launch_w1_noipra((int)a.reg_nv, d_args, st, sl, me, hr);   // (part 12)
#endif
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}
#endif


#if NPHIP_HAS(8)
// register-resident, one wave per chain, under the low-rank metric (settings.low_rank_metric; Machine<..., LR>): D <= 1024
hipError_t launch_fam_w1_lr(const Args& a, const Args* d_args, hipStream_t st, const LaunchSlice sl) {
    const dim3 g(((unsigned)sl.chain_n + 3) / 4), b(256);
    const int me = a.max_evals, hr = a.have_result;
    (void)g; (void)b; (void)me; (void)hr;
    switch (a.reg_nv) {
#if defined(NPHIP_DEV_W1NV_LR)
        case NPHIP_DEV_W1NV_LR: hipLaunchKernelGGL((k_advance<true, 1, NPHIP_DEV_W1NV_LR, false, false, true>), g, b, 0, st, d_args, me, hr, sl); break;
#elif !defined(NPHIP_DEV_BUILD)
        case 1: hipLaunchKernelGGL((k_advance<true, 1, 1, false, false, true>), g, b, 0, st, d_args, me, hr, sl); break;
        case 2: hipLaunchKernelGGL((k_advance<true, 1, 2, false, false, true>), g, b, 0, st, d_args, me, hr, sl); break;
        case 3: hipLaunchKernelGGL((k_advance<true, 1, 3, false, false, true>), g, b, 0, st, d_args, me, hr, sl); break;
        case 4: hipLaunchKernelGGL((k_advance<true, 1, 4, false, false, true>), g, b, 0, st, d_args, me, hr, sl); break;
        case 5: hipLaunchKernelGGL((k_advance<true, 1, 5, false, false, true>), g, b, 0, st, d_args, me, hr, sl); break;
        case 6: hipLaunchKernelGGL((k_advance<true, 1, 6, false, false, true>), g, b, 0, st, d_args, me, hr, sl); break;
        case 7: hipLaunchKernelGGL((k_advance<true, 1, 7, false, false, true>), g, b, 0, st, d_args, me, hr, sl); break;
        case 8: hipLaunchKernelGGL((k_advance<true, 1, 8, false, false, true>), g, b, 0, st, d_args, me, hr, sl); break;
#endif
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}
#endif

#if NPHIP_HAS(9)
// register-resident, two / four waves per chain, under the low-rank metric: the geometries choose_waves() picks (1024 < D <= 2048:
// two waves, 2048 < D <= 4096: four; 5..8 chunks per wave)
hipError_t launch_fam_rw_lr(const Args& a, const Args* d_args, int W, hipStream_t st, const LaunchSlice sl) {
#define NPHIP_LAUNCH_RW_LR(WW, NN) hipLaunchKernelGGL((k_advance<true, WW, NN, false, false, true>), g, b, 0, st, d_args, a.max_evals, a.have_result, sl)
    const dim3 g((unsigned)sl.chain_n), b(64 * W);
    (void)g; (void)b;
#if defined(NPHIP_DEV_RW_LR_W) && defined(NPHIP_DEV_RW_LR_NV)
    if (W != NPHIP_DEV_RW_LR_W || a.reg_nv != NPHIP_DEV_RW_LR_NV) return hipErrorInvalidValue;
    NPHIP_LAUNCH_RW_LR(NPHIP_DEV_RW_LR_W, NPHIP_DEV_RW_LR_NV);
    return hipGetLastError();
#elif defined(NPHIP_DEV_BUILD)
    return hipErrorInvalidValue;
#else
    if (W == 2) switch (a.reg_nv) {
        case 5: NPHIP_LAUNCH_RW_LR(2, 5); break;
        case 6: NPHIP_LAUNCH_RW_LR(2, 6); break;
        case 7: NPHIP_LAUNCH_RW_LR(2, 7); break;
        case 8: NPHIP_LAUNCH_RW_LR(2, 8); break;
        default: return hipErrorInvalidValue;
    } else switch (a.reg_nv) {
        case 5: NPHIP_LAUNCH_RW_LR(4, 5); break;
        case 6: NPHIP_LAUNCH_RW_LR(4, 6); break;
        case 7: NPHIP_LAUNCH_RW_LR(4, 7); break;
        case 8: NPHIP_LAUNCH_RW_LR(4, 8); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
#endif
#undef NPHIP_LAUNCH_RW_LR
}
#endif

#if NPHIP_HAS(4)
// memory-resident kernels: fused models of any D and W (D > 10 240, store_divergences, no_register_kernel) and the two-phase
// callback kernels
template <bool FUSED>
static hipError_t launch_mem_t(const Args& a, const Args* d_args, int W, hipStream_t st, const LaunchSlice sl) {
#if defined(NPHIP_DEV_CB_W)   // developer build: the callback kernel of ONE geometry
    if constexpr (!FUSED) {
        if (W == NPHIP_DEV_CB_W && !a.lr_on) {
            const unsigned n_ = (unsigned)sl.chain_n;
            hipLaunchKernelGGL((k_advance<false, NPHIP_DEV_CB_W, 0>), dim3(NPHIP_DEV_CB_W == 1 ? (n_ + 3) / 4 : n_), dim3(NPHIP_DEV_CB_W == 1 ? 256 : 64 * NPHIP_DEV_CB_W),
                               (NPHIP_DEV_CB_W >= 8 && a.sig_lds) ? (size_t)a.ld * 8 : 0, st, d_args, a.max_evals, a.have_result, sl);
            return hipGetLastError();
        }
    }
    return hipErrorInvalidValue;
#elif defined(NPHIP_DEV_BUILD)
    return hipErrorInvalidValue;
#else
    const unsigned n = (unsigned)sl.chain_n;
    const int me = a.max_evals, hr = a.have_result;
    if (FUSED && a.stream_cache && W == 1) {
        // memory-resident kernel with the cursor's (sigma^2, grad, p, rho) cached in VGPRs between leaves.  Measured
        // with more waves per chain (D > 1024) the cache costs occupancy or spills and does not pay; W == 1 only.
        hipLaunchKernelGGL((k_advance<true, 1, -8>), dim3((n + 3) / 4), dim3(256), 0, st, d_args, me, hr, sl);
        return hipGetLastError();
    }
    if (a.lr_on) switch (W) {   // the low-rank metric: the NV = -1 instantiations
        case 1: hipLaunchKernelGGL((k_advance<FUSED, 1, -1>), dim3((n + 3) / 4), dim3(256), 0, st, d_args, me, hr, sl); return hipGetLastError();
        case 2: hipLaunchKernelGGL((k_advance<FUSED, 2, -1>), dim3(n), dim3(128), 0, st, d_args, me, hr, sl); return hipGetLastError();
        case 4: hipLaunchKernelGGL((k_advance<FUSED, 4, -1>), dim3(n), dim3(256), 0, st, d_args, me, hr, sl); return hipGetLastError();
        case 8: hipLaunchKernelGGL((k_advance<FUSED, 8, -1>), dim3(n), dim3(512), a.sig_lds ? (size_t)a.ld * 8 : 0, st, d_args, me, hr, sl); return hipGetLastError();
        case 16: hipLaunchKernelGGL((k_advance<FUSED, 16, -1>), dim3(n), dim3(1024), a.sig_lds ? (size_t)a.ld * 8 : 0, st, d_args, me, hr, sl); return hipGetLastError();
        default: return hipErrorInvalidValue;
    }
    switch (W) {
        case 1: hipLaunchKernelGGL((k_advance<FUSED, 1, 0>), dim3((n + 3) / 4), dim3(256), 0, st, d_args, me, hr, sl); break;
        case 2: hipLaunchKernelGGL((k_advance<FUSED, 2, 0>), dim3(n), dim3(128), 0, st, d_args, me, hr, sl); break;
        case 4: hipLaunchKernelGGL((k_advance<FUSED, 4, 0>), dim3(n), dim3(256), 0, st, d_args, me, hr, sl); break;
        case 8: hipLaunchKernelGGL((k_advance<FUSED, 8, 0>), dim3(n), dim3(512), a.sig_lds ? (size_t)a.ld * 8 : 0, st, d_args, me, hr, sl); break;
        case 16: hipLaunchKernelGGL((k_advance<FUSED, 16, 0>), dim3(n), dim3(1024), a.sig_lds ? (size_t)a.ld * 8 : 0, st, d_args, me, hr, sl); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
#endif
}
hipError_t launch_fam_mem(const Args& a, const Args* d_args, bool fused, int W, hipStream_t st, const LaunchSlice sl) {
    return fused ? launch_mem_t<true>(a, d_args, W, st, sl) : launch_mem_t<false>(a, d_args, W, st, sl);
}
#endif

#if NPHIP_HAS(0)
// Resume chains stopped in PH_WAIT_HOST at new positions (host-driven re-parametrisation): the position goes to Q-pool buffer 0
// (and the callback staging row), the chain re-enters the initial-point sequence — evaluate, mass matrix from the gradient,
// step-size search — and continues with its next draw.  One block per resumed chain.
__global__ void k_resume(const Args* __restrict__ Ap, int n, const int64_t* __restrict__ chains, const double* __restrict__ pos, int fused) {
    const Args& A = *Ap;
    if ((int)blockIdx.x >= n) return;
    const int64_t ch = chains[blockIdx.x];
    Ctl* c = A.ctl + ch;
    if (c->phase != PH_WAIT_HOST) return;
    double* q = A.qpool + (size_t)ch * A.nqpool * 2 * A.ld;
    for (int64_t i = threadIdx.x; i < A.ld; i += blockDim.x) {
        const double v = i < A.dim ? pos[(size_t)blockIdx.x * A.dim + i] : 0.0;
        q[i] = v;
        if (!fused && i < A.dim) A.qeval[(size_t)ch * A.dim + i] = v;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        c->init_attempt = -1;   // (cont_init: fail instead of retrying elsewhere)
        c->eval_buf = 0;
        c->phase = PH_INIT_EVAL;
    }
}

// A new metric for chains stopped in PH_WAIT_HOST (host-driven low-rank adaptation): sigma^2 [n][dim], k rows of V [n][k][dim],
// lambda [n][k]; the chain keeps its position and goes on through a step-size search (PH_RESUME_SS).  One block per chain.
__global__ void k_set_metric(const Args* __restrict__ Ap, int n, const int64_t* __restrict__ chains, int k, const double* __restrict__ sig2,
                             const double* __restrict__ V, const double* __restrict__ lam, int* __restrict__ taken) {
    const Args& A = *Ap;
    if ((int)blockIdx.x >= n) return;
    const int64_t ch = chains[blockIdx.x];
    Ctl* c = A.ctl + ch;
    if (c->phase != PH_WAIT_HOST) return;
    for (int64_t i = threadIdx.x; i < A.ld; i += blockDim.x) {
        const double s2 = i < A.dim ? sig2[(size_t)blockIdx.x * A.dim + i] : 1.0;
        A.sig2[(size_t)ch * A.ld + i] = s2;
        A.lr_std[(size_t)ch * A.ld + i] = sqrt(s2);
        for (int j = 0; j < kLrMax; ++j)
            A.lr_V[((size_t)ch * kLrMax + j) * A.ld + i] = (j < k && i < A.dim) ? (float)V[((size_t)blockIdx.x * k + j) * A.dim + i] : 0.0f;   // (rounded to nearest: the contract)
    }
    if (threadIdx.x < kLrMax) A.lr_lam[(size_t)ch * kLrMax + threadIdx.x] = (int)threadIdx.x < k ? lam[(size_t)blockIdx.x * k + threadIdx.x] : 1.0;
    __syncthreads();
    if (threadIdx.x == 0) {
        c->host_metric = 1;
        c->lr_k = k;
        c->phase = PH_RESUME_SS;
        atomicAdd(taken, 1);   // (the host compares with n: a chain that was not stopped keeps its metric, and the caller is told)
    }
}

// The same metric for chains that RUN (nphip_sampler_stage_metric; between two launches): it is parked in Args::st_* and the chain takes
// it itself at the end of its current draw (Machine::apply_staged) — as long as it is still in its warm-up; a chain that is past it, has
// finished or failed is not counted in `taken`.  A metric that is still parked is replaced.
__global__ void k_stage_metric(const Args* __restrict__ Ap, int n, const int64_t* __restrict__ chains, int k, const double* __restrict__ sig2,
                               const double* __restrict__ V, const double* __restrict__ lam, int* __restrict__ taken) {
    const Args& A = *Ap;
    if ((int)blockIdx.x >= n) return;
    const int64_t ch = chains[blockIdx.x];
    Ctl* c = A.ctl + ch;
    if (c->phase == PH_DONE || c->phase == PH_ERROR || c->draw + 1 >= A.s.num_tune) return;
    for (int64_t i = threadIdx.x; i < A.ld; i += blockDim.x) {
        A.st_sig2[(size_t)ch * A.ld + i] = i < A.dim ? sig2[(size_t)blockIdx.x * A.dim + i] : 1.0;
        for (int j = 0; j < kLrMax; ++j)
            A.st_V[((size_t)ch * kLrMax + j) * A.ld + i] = (j < k && i < A.dim) ? (float)V[((size_t)blockIdx.x * k + j) * A.dim + i] : 0.0f;
    }
    if (threadIdx.x < kLrMax) A.st_lam[(size_t)ch * kLrMax + threadIdx.x] = (int)threadIdx.x < k ? lam[(size_t)blockIdx.x * k + threadIdx.x] : 1.0;
    __syncthreads();
    if (threadIdx.x == 0) {
        c->staged = k + 1;
        atomicAdd(taken, 1);
    }
}

hipError_t launch_stage_metric(const Args* d_args, int n, const int64_t* d_chains, int k, const double* sig2, const double* V, const double* lam, int* d_taken, hipStream_t st) {
    hipLaunchKernelGGL(k_stage_metric, dim3((unsigned)n), dim3(256), 0, st, d_args, n, d_chains, k, sig2, V, lam, d_taken);
    return hipGetLastError();
}

hipError_t launch_set_metric(const Args* d_args, int n, const int64_t* d_chains, int k, const double* sig2, const double* V, const double* lam, int* d_taken, hipStream_t st) {
    hipLaunchKernelGGL(k_set_metric, dim3((unsigned)n), dim3(256), 0, st, d_args, n, d_chains, k, sig2, V, lam, d_taken);
    return hipGetLastError();
}

hipError_t launch_resume(const Args* d_args, int n, const int64_t* d_chains, const double* d_pos, bool fused, hipStream_t st) {
    hipLaunchKernelGGL(k_resume, dim3((unsigned)n), dim3(256), 0, st, d_args, n, d_chains, d_pos, fused ? 1 : 0);
    return hipGetLastError();
}

hipError_t launch_advance(const Args& a, const Args* d_args, bool fused, int W, hipStream_t st, const LaunchSlice* slice) {
    LaunchSlice sl;
    if (slice) sl = *slice;
    else { sl.chain_lo = 0; sl.chain_n = (int)a.n_chains; sl.grp = -1; sl.seq = 0u; sl.materialise = 0; }
    sl.n_grp = 0;
    if (fused && a.lean && a.reg_nv > 0) return W == 4 ? launch_fam_lean4(a, d_args, st, sl) : (W == 8 ? launch_fam_lean8(a, d_args, st, sl) : hipErrorInvalidValue);
    if (fused && (W == 2 || W == 4) && a.reg_nv > 0) return a.lr_on ? launch_fam_rw_lr(a, d_args, W, st, sl) : launch_fam_rw(a, d_args, W, st, sl);
    if (fused && W == 1 && a.reg_nv > 0) return a.lr_on ? launch_fam_w1_lr(a, d_args, st, sl) : launch_fam_w1(a, d_args, st, sl);
    return launch_fam_mem(a, d_args, fused, W, st, sl);
}
#endif   // part 0

#if NPHIP_HAS(5)

// Resident launch of a host-callback job (k_advance<..., REMOTE>): `W` waves per chain with `nv` chunks of 128 elements each in
// registers (dim <= 128 W nv; W = 1: four chains per workgroup, W = 2 / 4: one).  The slice names the groups and their
// sequence numbers.
hipError_t launch_remote_w1(const Args* d_args, int nv, hipStream_t st, const LaunchSlice& sl);     // part 5
hipError_t launch_remote_wn(const Args* d_args, int W, int nv, hipStream_t st, const LaunchSlice& sl);   // part 6
#define NPHIP_LAUNCH_REMOTE(WW, NN) hipLaunchKernelGGL((k_advance<false, WW, NN, false, true>), g, b, 0, st, d_args, 0, 0, sl)
hipError_t launch_remote_w1(const Args* d_args, int nv, hipStream_t st, const LaunchSlice& sl) {
#ifdef NPHIP_DEV_BUILD
    return hipErrorInvalidValue;
#else
    const dim3 g(((unsigned)sl.chain_n + 3) / 4), b(256);
    switch (nv) {
        case 1: NPHIP_LAUNCH_REMOTE(1, 1); break;
        case 2: NPHIP_LAUNCH_REMOTE(1, 2); break;
        case 3: NPHIP_LAUNCH_REMOTE(1, 3); break;
        case 4: NPHIP_LAUNCH_REMOTE(1, 4); break;
        case 5: NPHIP_LAUNCH_REMOTE(1, 5); break;
        case 6: NPHIP_LAUNCH_REMOTE(1, 6); break;
        case 7: NPHIP_LAUNCH_REMOTE(1, 7); break;
        case 8: NPHIP_LAUNCH_REMOTE(1, 8); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
#endif
}
hipError_t launch_remote(const Args* d_args, int W, int nv, hipStream_t st, const LaunchSlice& sl) {
    return W == 1 ? launch_remote_w1(d_args, nv, st, sl) : launch_remote_wn(d_args, W, nv, st, sl);
}
#undef NPHIP_LAUNCH_REMOTE

#endif   // part 5

#if NPHIP_HAS(6)
#define NPHIP_LAUNCH_REMOTE(WW, NN) hipLaunchKernelGGL((k_advance<false, WW, NN, false, true>), g, b, 0, st, d_args, 0, 0, sl)
hipError_t launch_remote_wn(const Args* d_args, int W, int nv, hipStream_t st, const LaunchSlice& sl) {
#ifdef NPHIP_DEV_BUILD
    return hipErrorInvalidValue;
#else
    const dim3 g((unsigned)sl.chain_n), b(64 * W);
    if (W == 2) switch (nv) {
        case 1: NPHIP_LAUNCH_REMOTE(2, 1); break;
        case 2: NPHIP_LAUNCH_REMOTE(2, 2); break;
        case 3: NPHIP_LAUNCH_REMOTE(2, 3); break;
        case 4: NPHIP_LAUNCH_REMOTE(2, 4); break;
        case 5: NPHIP_LAUNCH_REMOTE(2, 5); break;
        case 6: NPHIP_LAUNCH_REMOTE(2, 6); break;
        case 7: NPHIP_LAUNCH_REMOTE(2, 7); break;
        case 8: NPHIP_LAUNCH_REMOTE(2, 8); break;
        default: return hipErrorInvalidValue;
    } else if (W == 4) switch (nv) {
        case 1: NPHIP_LAUNCH_REMOTE(4, 1); break;
        case 2: NPHIP_LAUNCH_REMOTE(4, 2); break;
        case 3: NPHIP_LAUNCH_REMOTE(4, 3); break;
        case 4: NPHIP_LAUNCH_REMOTE(4, 4); break;
        case 5: NPHIP_LAUNCH_REMOTE(4, 5); break;
        case 6: NPHIP_LAUNCH_REMOTE(4, 6); break;
        case 7: NPHIP_LAUNCH_REMOTE(4, 7); break;
        case 8: NPHIP_LAUNCH_REMOTE(4, 8); break;
        default: return hipErrorInvalidValue;
    } else return hipErrorInvalidValue;
    return hipGetLastError();
#endif
}
#undef NPHIP_LAUNCH_REMOTE
#endif   // part 6


#if NPHIP_HAS(11)
// Resident launch of the dense-precision Gaussian (k_advance<..., REMOTE, DENSEG>): one wave per chain, `nv` chunks of 128 dimensions,
// every chain of the job on the device at once (<= 1024: one workgroup of four chains per CU) — the launch's roll call makes sure.
hipError_t launch_dense_resident(const Args* d_args, int nv, int max_evals, hipStream_t st, const LaunchSlice sl) {
    const dim3 g((unsigned)((sl.chain_n + 3) / 4)), b(256);
#define NPHIP_LAUNCH_DG(NN) hipLaunchKernelGGL((k_advance<false, 1, NN, false, true, false, false, true>), g, b, 0, st, d_args, max_evals, 0, sl)
    switch (nv) {
#ifdef NPHIP_DEV_DG_NV
        case NPHIP_DEV_DG_NV: NPHIP_LAUNCH_DG(NPHIP_DEV_DG_NV); break;
#else
        case 1: NPHIP_LAUNCH_DG(1); break;
        case 2: NPHIP_LAUNCH_DG(2); break;
        case 3: NPHIP_LAUNCH_DG(3); break;
        case 4: NPHIP_LAUNCH_DG(4); break;
        case 5: NPHIP_LAUNCH_DG(5); break;
        case 6: NPHIP_LAUNCH_DG(6); break;
        case 7: NPHIP_LAUNCH_DG(7); break;
        case 8: NPHIP_LAUNCH_DG(8); break;
#endif
        default: return hipErrorInvalidValue;
    }
#undef NPHIP_LAUNCH_DG
    return hipGetLastError();
}
#endif   // part 11

#if NPHIP_PART == 7
// ----------------------------------------------------------------------------------------
// The runtime-compiled part (one model's density; nutpie_amd/density.py builds it with -DNPHIP_JIT_DENSITY -DNPHIP_PART=7
// -DNPHIP_JIT_NV=<chunks of 128 dimensions>): the resident kernel with the density called inside the leaf, and the same density
// as a plain batched kernel (the launch-per-evaluation form, nphip_device_logp_fn: any dimension, store_divergences, the
// low-rank wrapper ...).  Both evaluate nphip_density with one wavefront per chain.
// ----------------------------------------------------------------------------------------
#if !NPHIP_JIT || !defined(NPHIP_JIT_NV)
#error "part 7 is the runtime-compiled density: -DNPHIP_JIT_DENSITY -DNPHIP_JIT_NV=n behind a prelude that defines NphipData / nphip_density"
#endif
#ifndef NPHIP_JIT_W
#define NPHIP_JIT_W 1
#endif
// -DNPHIP_JIT_LR=1: the library's resident kernel integrates under the low-rank metric (Machine<..., LR>)
#ifndef NPHIP_JIT_LR
#define NPHIP_JIT_LR 0
#endif
// one wave per chain: four chains per workgroup; NPHIP_JIT_W waves per chain: one chain per workgroup of 64 W threads
__global__ __launch_bounds__(NPHIP_JIT_W == 1 ? 256 : 64 * NPHIP_JIT_W) void k_density_batch(const NphipData* __restrict__ data, uint64_t n_chains, int dim, const double* __restrict__ q,
                                                       double* __restrict__ grad, double* __restrict__ logp, int lds_doubles, int shared_doubles, int rows_in_lds) {
    extern __shared__ __attribute__((aligned(16))) double s_scratch[];
    constexpr int CPB = NPHIP_JIT_W == 1 ? 4 : 1;   // chains per block
    constexpr int TPC = 64 * NPHIP_JIT_W;           // threads per chain
    const int slot = NPHIP_JIT_W == 1 ? (int)(threadIdx.x >> 6) : 0, tid = NPHIP_JIT_W == 1 ? (int)(threadIdx.x & 63) : (int)threadIdx.x;
    const uint64_t chain = (uint64_t)blockIdx.x * CPB + slot;
    double* shared = s_scratch + (size_t)CPB * lds_doubles;
    // the chain's position and gradient rows in LDS, as in the resident kernel (the density reads x[] several times)
    const int ldp = (dim + 1) & ~1;
    double* rows = shared + shared_doubles + (size_t)slot * 2 * ldp;
    if (threadIdx.x == 0) nphip_chains_per_block_ = CPB;
    nphip_density_stage(*data, shared, (int)threadIdx.x, (int)blockDim.x);
    if (rows_in_lds && chain < n_chains) for (int i = tid; i < dim; i += TPC) rows[i] = q[chain * (uint64_t)dim + i];
    __syncthreads();
    if (chain >= n_chains) return;
    if (!rows_in_lds) {   // (rows too long for LDS: the density works on the rows in memory)
        const double lp = nphip_density(*data, dim, q + chain * (uint64_t)dim, grad + chain * (uint64_t)dim, s_scratch + (size_t)slot * lds_doubles, shared, tid);
        if (tid == 0) logp[chain] = lp;
        return;
    }
    const double lp = nphip_density(*data, dim, rows, rows + ldp, s_scratch + (size_t)slot * lds_doubles, shared, tid);
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    if (NPHIP_JIT_W == 1) __builtin_amdgcn_wave_barrier(); else __syncthreads();
    for (int i = tid; i < dim; i += TPC) grad[chain * (uint64_t)dim + i] = rows[ldp + i];
    if (tid == 0) logp[chain] = lp;
}
}  // namespace nphip
extern "C" {
// nphip_jit_launch_fn (host.hip): one launch of the resident kernel over the slice's chains
int nphip_jit_launch(const nphip::Args* d_args, int max_evals, void* stream, const nphip::LaunchSlice* sl, uint64_t dyn_lds_bytes) {
    // one wave per chain: four chains per workgroup (every SIMD of a CU), or ONE when the job has no more chains than the device has
    // CUs: the chains of a workgroup share its LDS bandwidth (256 chains: +9-12 %, profiles/r5_jit_chains_per_workgroup.txt).  Not two
    // per workgroup up to 512 chains: +1-3 % for that job, but its 256 workgroups then hold every CU's LDS and a second job of the
    // same size no longer runs beside it (two concurrent 512-chain jobs: 70 instead of ~115 M leapfrogs/s).  (NPHIP_JIT_CPB=1|2|4
    // overrides: measurements)
    static const int forced = [] { const char* e = getenv("NPHIP_JIT_CPB"); const int v = e ? atoi(e) : 0; return (v == 1 || v == 2 || v == 4) ? v : 0; }();
    nphip::LaunchSlice s2 = *sl;
    s2.cpb = forced ? forced : (sl->chain_n > 256 ? 4 : 1);
    const dim3 g(NPHIP_JIT_W == 1 ? ((unsigned)sl->chain_n + s2.cpb - 1) / s2.cpb : (unsigned)sl->chain_n), b(NPHIP_JIT_W == 1 ? 256 : 64 * NPHIP_JIT_W);
    hipLaunchKernelGGL((nphip::k_advance<false, NPHIP_JIT_W, NPHIP_JIT_NV, false, true, (NPHIP_JIT_LR != 0)>), g, b, (size_t)dyn_lds_bytes, (hipStream_t)stream, d_args, max_evals, 0, s2);
    return (int)hipGetLastError();
}
int nphip_jit_w(void) { return NPHIP_JIT_W; }
int nphip_jit_lr(void) { return NPHIP_JIT_LR; }
int nphip_jit_nv(void) { return NPHIP_JIT_NV; }
// nphip_device_logp_fn; user_data -> { device pointer of the data block, LDS doubles per wave }
struct nphip_jit_batch_t { const void* data; int32_t lds_doubles, shared_doubles; };
int nphip_jit_logp(uint64_t n_chains, uint64_t dim, const double* q, double* grad, double* logp, void* stream, void* user_data) {
    const nphip_jit_batch_t* u = (const nphip_jit_batch_t*)user_data;
    if (!u) return -1;
    constexpr int CPB = NPHIP_JIT_W == 1 ? 4 : 1;
    const size_t own = (size_t)CPB * u->lds_doubles + u->shared_doubles, rows = (size_t)CPB * 2 * (((size_t)dim + 1) & ~(size_t)1);
    const int rows_in_lds = (own + rows) * sizeof(double) <= 144 * 1024;
    hipLaunchKernelGGL(nphip::k_density_batch, dim3((unsigned)((n_chains + CPB - 1) / CPB)), dim3(NPHIP_JIT_W == 1 ? 256 : 64 * NPHIP_JIT_W),
                       (own + (rows_in_lds ? rows : 0)) * sizeof(double), (hipStream_t)stream,
                       (const NphipData*)u->data, n_chains, (int)dim, q, grad, logp, (int)u->lds_doubles, (int)u->shared_doubles, rows_in_lds);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
int nphip_jit_has_expand(void) {
#ifdef NPHIP_JIT_EXPAND
    return 1;
#else
    return 0;
#endif
}
#ifdef NPHIP_JIT_EXPAND
}  // extern "C"
namespace nphip {
// The model's expand step (nphip_expand, generated with the density: constrained parameters and deterministic values of ONE draw)
// over a block of stored draws: one wavefront (NPHIP_JIT_W wavefronts) per row, as the density is evaluated.
__global__ __launch_bounds__(NPHIP_JIT_W == 1 ? 256 : 64 * NPHIP_JIT_W) void k_expand_batch(const NphipData* __restrict__ data, uint64_t n_rows, int dim, uint64_t expanded,
                                                      const double* __restrict__ x, double* __restrict__ out, int lds_doubles, int shared_doubles) {
    extern __shared__ __attribute__((aligned(16))) double s_scratch[];
    constexpr int CPB = NPHIP_JIT_W == 1 ? 4 : 1;
    const int slot = NPHIP_JIT_W == 1 ? (int)(threadIdx.x >> 6) : 0, tid = NPHIP_JIT_W == 1 ? (int)(threadIdx.x & 63) : (int)threadIdx.x;
    const uint64_t row = (uint64_t)blockIdx.x * CPB + slot;
    double* shared = s_scratch + (size_t)CPB * lds_doubles;
    if (threadIdx.x == 0) nphip_chains_per_block_ = CPB;
    nphip_density_stage(*data, shared, (int)threadIdx.x, (int)blockDim.x);
    __syncthreads();
    if (row >= n_rows) return;
    (void)nphip_expand(*data, dim, x + row * (uint64_t)dim, out + row * expanded, s_scratch + (size_t)slot * lds_doubles, shared, tid);
}
}  // namespace nphip
extern "C" {
// nphip_device_expand_fn (include/nutpie_hip.h); user_data -> nphip_jit_batch_t with the expand function's LDS need
int nphip_jit_expand(uint64_t n_rows, uint64_t dim, uint64_t expanded_dim, const double* x, double* out, void* stream, void* user_data) {
    const nphip_jit_batch_t* u = (const nphip_jit_batch_t*)user_data;
    if (!u) return -1;
    if (n_rows == 0) return 0;
    constexpr int CPB = NPHIP_JIT_W == 1 ? 4 : 1;
    const size_t own = (size_t)CPB * u->lds_doubles + u->shared_doubles;
    hipLaunchKernelGGL(nphip::k_expand_batch, dim3((unsigned)((n_rows + CPB - 1) / CPB)), dim3(NPHIP_JIT_W == 1 ? 256 : 64 * NPHIP_JIT_W), own * sizeof(double),
                       (hipStream_t)stream, (const NphipData*)u->data, n_rows, (int)dim, expanded_dim, x, out, (int)u->lds_doubles, (int)u->shared_doubles);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
#else
int nphip_jit_expand(uint64_t, uint64_t, uint64_t, const double*, double*, void*, void*) { return -1; }
#endif
}  // extern "C"
namespace nphip {
#endif   // part 7

#if NPHIP_HAS(10)
// ----------------------------------------------------------------------------------------
// Dense-precision Gaussian (nphip_model_dense_gaussian; BASELINE.json configs[1] variant (ii)): the gradient of ALL chains as one
// fp64 GEMM on the matrix cores, hand-written (dense_tile.h) — the only place in the engine where MFMA applies: 2 D^2 flop per
// chain and evaluation against the leapfrog's 10 D.  These two kernels are the launch-per-evaluation form (the model behind the
// engine's own device-callback path: any number of chains, any dimension); the resident form calls the same tile from inside the
// register-resident leaf (Machine<..., DENSEG>).
// ----------------------------------------------------------------------------------------
// G[chain][j] = -sum_k (X[chain][k] - mu[k]) Pp[j][k] for a 64 x 64 tile per workgroup (four waves, 32 x 32 each).  X, G: dense
// [n][D]; Pp: [DP][KP] (DP = D rounded up to 64, KP to 16), zero-padded; mu: [KP].
// Tile order: workgroup b runs on XCD b % 8 (round-robin dispatch); the tiles are dealt so that an XCD owns a contiguous run of
// them — neighbouring tiles share their rows of X or of P in that XCD's L2.
__global__ __launch_bounds__(256) void k_dense_grad(const double* __restrict__ X, const double* __restrict__ Pp, const double* __restrict__ mu,
                                                    double* __restrict__ G, int64_t n, int64_t D, int64_t KP) {
    const int64_t Mt = (n + 63) / 64, Nt = (D + 63) / 64, T = Mt * Nt, tpx = (T + 7) / 8;
    const int64_t b = blockIdx.x, t = (b & 7) * tpx + (b >> 3);
    if ((b >> 3) >= tpx || t >= T) return;
    const int64_t mt = t / Nt, nt = t % Nt;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t r0 = mt * 64 + (wave >> 1) * 32, c0 = nt * 64 + (wave & 1) * 32;
    if (r0 >= n || c0 >= D) return;   // (whole block of this wave outside the matrix)
    const double* xrow[2];
    const double* prow[2];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        int64_t i = r0 + 16 * r + (lane & 15);
        i = i < n ? i : n - 1;    // (rows past the end: the last row again, never stored)
        xrow[r] = X + (size_t)i * D;
        prow[r] = Pp + (size_t)(c0 + 16 * r + (lane & 15)) * KP;
    }
    dg_v4 acc[2][2];
    dense_block_32x32(xrow, prow, mu, D, lane, acc);
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int64_t j = c0 + 16 * c + (lane & 15);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int64_t i = r0 + 16 * r + (lane >> 4) + 4 * q;
                if (i < n && j < D) st1(G, i * D + j, -acc[r][c][q]);
            }
        }
}

// logp[chain] = 1/2 sum_i z_i g_i (z = x - mu) in the engine's summation order with W waves per chain (include/nphip_spec.h)
template <int W>
__global__ __launch_bounds__(64 * W) void k_dense_logp(const double* __restrict__ X, const double* __restrict__ mu, const double* __restrict__ G,
                                                       double* __restrict__ logp, int64_t n, int64_t D) {
    __shared__ double red_[8 * W];
    LdsDouble red = (LdsDouble)red_;
    const int64_t chain = blockIdx.x;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t nch = (D + 127) / 128;
    const double* x = X + (size_t)chain * D;
    const double* g = G + (size_t)chain * D;
    double2 acc = {0.0, 0.0};
    for (int64_t cc = wave; cc < nch; cc += W) {
        const int64_t i = cc * 128 + 2 * lane;
        const double2 xv = ld2_dense(x, i, D), gv = ld2_dense(g, i, D);
        const double zx = xv.x - ((i < D) ? ld1(mu, i) : 0.0), zy = xv.y - ((i + 1 < D) ? ld1(mu, i + 1) : 0.0);
        acc.x = fma(zx, gv.x, acc.x);
        acc.y = fma(zy, gv.y, acc.y);
    }
    double a = acc.x + acc.y, bdummy = 0.0;
    reduce2<W>(a, bdummy, red);
    if (threadIdx.x == 0) logp[chain] = 0.5 * a;
}

// fp64 matrix-core issue rate, measured: every wave issues `iters` x 8 independent-accumulator v_mfma_f64_16x16x4_f64 back to back (no memory
// traffic); the host divides the flops by the kernel's duration.  What bench.py prices the dense model's `mfma` roofline against, beside
// the datasheet figure.
__global__ __launch_bounds__(256) void k_mfma_f64_rate(double* out, int iters) {
    dg_v4 acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = (dg_v4){0.0, 0.0, 0.0, 0.0};
    double a = 1.0 + 1e-9 * (double)threadIdx.x, b = 1.0 - 1e-9 * (double)threadIdx.x;
    const long long c0 = (long long)__builtin_readcyclecounter(), w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    double t = 0.0;
#pragma unroll
    for (int i = 0; i < 8; ++i) t += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (t == 12345.678) out[0] = t;   // (keeps the accumulators alive)
    if (blockIdx.x == 0 && threadIdx.x == 0) {   // shader cycles and 100 MHz ticks of this wave's loop: cycles per MFMA and the clock it ran at
        out[1] = (double)((long long)__builtin_readcyclecounter() - c0);
        out[2] = (double)(wall_clock64() - w0);
    }
}
hipError_t launch_mfma_f64_rate(double* out, int blocks, int iters, hipStream_t st) {
    hipLaunchKernelGGL(k_mfma_f64_rate, dim3((unsigned)blocks), dim3(256), 0, st, out, iters);
    return hipGetLastError();
}

hipError_t launch_dense_grad(const double* X, const double* Pp, const double* mu, double* G, double* logp, int64_t n, int64_t D, int64_t KP, int W,
                             hipStream_t st) {
    const int64_t Mt = (n + 63) / 64, Nt = (D + 63) / 64, T = Mt * Nt, tpx = (T + 7) / 8;
    hipLaunchKernelGGL(k_dense_grad, dim3((unsigned)(8 * tpx)), dim3(256), 0, st, X, Pp, mu, G, n, D, KP);
    if (logp) {
        switch (W) {
            case 1: hipLaunchKernelGGL(k_dense_logp<1>, dim3((unsigned)n), dim3(64), 0, st, X, mu, G, logp, n, D); break;
            case 2: hipLaunchKernelGGL(k_dense_logp<2>, dim3((unsigned)n), dim3(128), 0, st, X, mu, G, logp, n, D); break;
            case 4: hipLaunchKernelGGL(k_dense_logp<4>, dim3((unsigned)n), dim3(256), 0, st, X, mu, G, logp, n, D); break;
            case 8: hipLaunchKernelGGL(k_dense_logp<8>, dim3((unsigned)n), dim3(512), 0, st, X, mu, G, logp, n, D); break;
            case 16: hipLaunchKernelGGL(k_dense_logp<16>, dim3((unsigned)n), dim3(1024), 0, st, X, mu, G, logp, n, D); break;
            default: return hipErrorInvalidValue;
        }
    }
    return hipGetLastError();
}
#endif   // part 10

#if NPHIP_HAS(0)
// ----------------------------------------------------------------------------------------
// test hooks: device implementations of the nphip_spec.h contract
// ----------------------------------------------------------------------------------------
__global__ void k_test_detmath(int fn, uint64_t n, const double* x, double* y) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (fn == 7) {
        const uint64_t seed = (uint64_t)x[0];
        const uint32_t chain = (uint32_t)x[1], draw = (uint32_t)x[2], purpose = (uint32_t)x[3];
        if (2 * i < n) {
            double z0, z1;
            nphip_normal_pair(nphip_philox(seed, (uint32_t)i, chain, draw, purpose), &z0, &z1);
            y[2 * i] = z0;
            if (2 * i + 1 < n) y[2 * i + 1] = z1;
        }
        return;
    }
    if (i >= n) return;
    double s, cs;
    switch (fn) {
        case 0: y[i] = nphip_exp(x[i]); break;
        case 1: y[i] = nphip_log(x[i]); break;
        case 2: y[i] = nphip_log1p(x[i]); break;
        case 3: nphip_sincos2pi(x[i], &s, &cs); y[i] = s; break;
        case 4: nphip_sincos2pi(x[i], &s, &cs); y[i] = cs; break;
        case 5: y[i] = sqrt(x[i]); break;
        default: y[i] = 1.0 / x[i]; break;
    }
}

template <int W>
__global__ void k_test_dot(uint64_t n, const double* x, const double* y, double* out) {
    __shared__ double red_[8 * W];
    LdsDouble red = (LdsDouble)red_;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t nch = (int64_t)((n + 127) / 128);
    double2 acc = {0.0, 0.0};
    for (int64_t cc = wave; cc < nch; cc += W) {
        int64_t i = cc * 128 + 2 * lane;
        double2 a = ld2_dense(x, i, (int64_t)n), b = ld2_dense(y, i, (int64_t)n);
        acc.x = fma(a.x, b.x, acc.x);
        acc.y = fma(a.y, b.y, acc.y);
    }
    double a = acc.x + acc.y, b = 0.0;
    reduce2<W>(a, b, red);
    if (threadIdx.x == 0) *out = a;
}

hipError_t launch_test_detmath(int fn, uint64_t n, const double* x, double* y, hipStream_t st) {
    uint64_t work = (fn == 7) ? (n + 1) / 2 : n;
    hipLaunchKernelGGL(k_test_detmath, dim3((unsigned)((work + 255) / 256)), dim3(256), 0, st, fn, n, x, y);
    return hipGetLastError();
}

hipError_t launch_test_dot(int W, uint64_t n, const double* x, const double* y, double* out, hipStream_t st) {
    switch (W) {
        case 1: hipLaunchKernelGGL(k_test_dot<1>, dim3(1), dim3(64), 0, st, n, x, y, out); break;
        case 2: hipLaunchKernelGGL(k_test_dot<2>, dim3(1), dim3(128), 0, st, n, x, y, out); break;
        case 4: hipLaunchKernelGGL(k_test_dot<4>, dim3(1), dim3(256), 0, st, n, x, y, out); break;
        case 8: hipLaunchKernelGGL(k_test_dot<8>, dim3(1), dim3(512), 0, st, n, x, y, out); break;
        case 16: hipLaunchKernelGGL(k_test_dot<16>, dim3(1), dim3(1024), 0, st, n, x, y, out); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}
#endif   // part 0

}  // namespace nphip
