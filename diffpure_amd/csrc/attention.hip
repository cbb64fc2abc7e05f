// Fused self-attention  out = softmax(q k^T / sqrt(d)) v  on the fp16 matrix cores with fp32-class accuracy
// (the "f16x3" arithmetic of igemm_h2.hip: operands split into fp16 hi + lo, three MFMA passes per product).
// Replaces, on the inference path, the QK^T GEMM + row softmax + PV GEMM trio (QKVAttentionLegacy.forward,
// /root/reference/guided_diffusion/unet.py:345-362; AttnBlockpp's einsums, score_sde/models/layerspp.py:82-86):
// the [B*heads, T, T] score tensor (2.1 GB per layer at 32x32, B=64) is never written.
//
// Two kernels:
//   attn_pack : qkv [B,T,3C] fp32 -> Q (pre-scaled by 1/sqrt(d)), K as h2 rows [z][T][d], and V TRANSPOSED per
//               32-key block, [z][T/32][d][32 keys], h2 along the key axis, keys stored in the order the
//               MFMA accumulator layout hands the probabilities back (see below).     z = b * heads + h
//   attn_flash: one workgroup = QW waves x 32 queries of one z; K / V^T blocks of 32 keys stream through LDS by
//               LDS-DMA (double buffered); per block and wave:
//                 S^T (32 keys x 32 queries) = K_blk Q^T        12 MFMA   (A = K rows, B = Q held in registers)
//                 online softmax: the C layout gives every lane ONE query (column lane & 31) and 16 of its keys, so
//                 the running max / sum / rescale are per-lane scalars plus one exchange with lane ^ 32
//                 O^T (d x 32 queries) += V^T_blk P^T           12 MFMA   (A = V^T rows, B = P straight from the
//                 accumulator registers - no shuffle: the key order inside a register octet is
//                 {0,1,2,3,8,9,10,11 | 4,5,6,7,12,13,14,15}, and attn_pack stores V^T in exactly that order)
// LDS rows are 128 bytes (32 k-values as hi|lo octets) with the XOR swizzle of igemm_h2.hip.
//
// ONE-PASS form (H1, round 4: the fp16 x fp16 precision modes - the arithmetic of the reference's own use_fp16 attention,
// unet.py:358-361): qkv arrives as plain fp16 (the qkv convolution stores it so), ONE MFMA pass per product.  Q and K are read
// where the convolution left them - a K slice of a key is 64 contiguous bytes of its qkv row, which is all an LDS-DMA lane needs -
// so attn_pack shrinks to the transposition of V (attn_pack_vt); the 1/sqrt(d) goes onto the scores.  LDS rows are 64 bytes
// (32 k-values) with the swizzle of igemm_h2_dw.hip: half the bytes through L2 -> LDS of a kernel that re-reads K and V T/128 times.
#include "dp_common.h"
#include "dp_tune.h"

namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));

// D = head dimension: 64 (guided diffusion: num_head_channels = 64) or 256 (NCSN++ AttnBlockpp at 16x16: one head of C = 256
// channels, layerspp.py:75-91 - in round 2 still two fp32 GEMMs and a softmax pass over a materialised [B, 256, 256] score
// tensor).  SL = D / 32 = 128-byte slices per K row.  With D = 256 a wave keeps Q (128 registers) and O^T (128 accumulators)
// resident: one wave per SIMD, 64 KB of K / V^T per stage.
constexpr int KB = 32;                // keys per block

__device__ __forceinline__ int swz128(int row, int slot) { return row * 128 + ((slot ^ ((row >> 1) & 7)) << 4); }
__device__ __forceinline__ int swz64(int row, int slot) { return row * 64 + ((slot ^ ((row >> 2) & 3)) << 4); }       // 64-byte rows: 4 slots

struct PackArgs {
    const float* qkv;
    int B, T, C, NH;
    int oq, ok, ov, sh;     // channel offsets of q, k, v of head 0 and the head stride
    float qscale;
    char* qh;               // [z][T][D*4]
    char* kh;               // [z][T][D*4]
    char* vt;               // [z][T/32][D][128]
};

__device__ __forceinline__ void split8(const float* v, float scale, half8& hi, half8& lo) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float x = v[j] * scale;
        hi[j] = (_Float16)x;
        lo[j] = (_Float16)(x - (float)hi[j]);
    }
}

// grid.x = z * (T/32) + block; 256 threads.  Q/K: thread -> (token, d-octet); V^T: thread -> (d, position octet)
template <int D>
__global__ __launch_bounds__(256) void attn_pack_kernel(PackArgs p) {
    const int nblk = p.T / KB;
    const int z = blockIdx.x / nblk, blk = blockIdx.x - z * nblk;
    const int b = z / p.NH, h = z - b * p.NH;
    const int tid = threadIdx.x;
    const size_t row0 = (size_t)b * p.T + (size_t)blk * KB;      // first token of this block in qkv
    const int c3 = 3 * p.C;
    for (int item = tid; item < KB * (D / 8); item += 256) {   // Q and K: 32 tokens x D/8 octets
        const int tok = item / (D / 8), oc = item - tok * (D / 8);
        const float* src = p.qkv + (row0 + tok) * c3 + h * p.sh + oc * 8;
        float v[8];
        half8 hi, lo;
        *reinterpret_cast<f32x4*>(v) = *reinterpret_cast<const f32x4*>(src + p.oq);
        *reinterpret_cast<f32x4*>(v + 4) = *reinterpret_cast<const f32x4*>(src + p.oq + 4);
        split8(v, p.qscale, hi, lo);
        half8* dq = reinterpret_cast<half8*>(p.qh + (((size_t)z * p.T + blk * KB + tok) * D + oc * 8) * 4);
        dq[0] = hi;
        dq[1] = lo;
        *reinterpret_cast<f32x4*>(v) = *reinterpret_cast<const f32x4*>(src + p.ok);
        *reinterpret_cast<f32x4*>(v + 4) = *reinterpret_cast<const f32x4*>(src + p.ok + 4);
        split8(v, 1.f, hi, lo);
        half8* dk = reinterpret_cast<half8*>(p.kh + (((size_t)z * p.T + blk * KB + tok) * D + oc * 8) * 4);
        dk[0] = hi;
        dk[1] = lo;
    }
    for (int item = tid; item < D * 4; item += 256) {   // V^T: D d x 4 position octets; lanes run along d (coalesced reads of V rows)
        const int dd = item % D, po = item / D;
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            // the key whose position is po*8 + j
            const int pos = po * 8 + j;
            const int p16 = pos & 15;
            const int kk = (pos & 16) + ((p16 >> 3) & 1) * 4 + ((p16 >> 2) & 1) * 8 + (p16 & 3);   // inverse of key_pos
            v[j] = p.qkv[(row0 + kk) * c3 + h * p.sh + p.ov + dd];
        }
        half8 hi, lo;
        split8(v, 1.f, hi, lo);
        half8* dv = reinterpret_cast<half8*>(p.vt + (((size_t)z * nblk + blk) * D + dd) * 128 + po * 32);
        dv[0] = hi;
        dv[1] = lo;
    }
}

// One-pass form: V^T only.  qkv16 [B][T][3C] fp16 -> vt [z][T/32][D][32 keys] fp16 (64-byte rows), keys in the accumulator order.
struct PackVtArgs {
    const _Float16* qkv;
    int T, C, NH;
    int ov, sh;             // channel offset of v of head 0 and the head stride
    char* vt;
};

template <int D>
__global__ __launch_bounds__(256) void attn_pack_vt_kernel(PackVtArgs p) {
    const int nblk = p.T / KB;
    const int z = blockIdx.x / nblk, blk = blockIdx.x - z * nblk;
    const int b = z / p.NH, h = z - b * p.NH;
    const size_t row0 = (size_t)b * p.T + (size_t)blk * KB;
    const int c3 = 3 * p.C;
    for (int item = threadIdx.x; item < D * 4; item += 256) {     // D d x 4 position octets; lanes run along d (coalesced reads of V rows)
        const int dd = item % D, po = item / D;
        half8 v;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int pos = po * 8 + j;
            const int p16 = pos & 15;
            const int kk = (pos & 16) + ((p16 >> 3) & 1) * 4 + ((p16 >> 2) & 1) * 8 + (p16 & 3);   // inverse of key_pos
            v[j] = p.qkv[(row0 + kk) * c3 + h * p.sh + p.ov + dd];
        }
        *reinterpret_cast<half8*>(p.vt + (((size_t)z * nblk + blk) * D + dd) * 64 + po * 16) = v;
    }
}

struct FlashArgs {
    const char* qh;
    const char* kh;
    const char* vt;
    float* out;         // [B][T][C] fp32, or (out16) the zero-bordered "h1" operand [B][T/W + 2][W + 2][C] fp16 (border pre-zeroed)
    int T, C, NH;
    int out16, W;
    // one-pass form: qh / kh point at q / k of head 0 inside qkv16 (fp16), rows of row_bytes, heads head_bytes apart; qscale on the scores
    int row_bytes, head_bytes;
    float qscale;
    int xcd_map;        // round 6 (DP_XCD_MAP): the query blocks of one (sample, head) on ONE XCD - they all stream the same K / V^T
};

#define AT_GLDS(src, dst)                                                                      \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src),     \
                                     (__attribute__((address_space(3))) void*)(dst), 16, 0, 0)

// QW waves per workgroup, 32 queries each.  LDS per stage: K tile SL slices x 32 rows x ROWB bytes, V^T tile D rows x ROWB bytes;
// ROWB = 128 (hi|lo octets, three passes) or 64 (H1: plain fp16, one pass).
template <int QW, int D, bool H1>
__global__ __launch_bounds__(QW * 64) __attribute__((amdgpu_waves_per_eu(1, D > 64 ? 1 : 8))) void attn_flash_kernel(FlashArgs p) {
    constexpr int NT = QW * 64;
    constexpr int SL = D / 32;
    constexpr int ROWB = H1 ? 64 : 128, SPR = ROWB / 16;                    // bytes / 16-byte slots per LDS row
    constexpr int KT = SL * KB * ROWB, VT = D * ROWB, STAGE = KT + VT;      // three-pass D = 64: 8 KB + 8 KB; D = 256: 32 KB + 32 KB
    constexpr int PIECES = STAGE / (NT * 16);                               // DMA instructions per thread per block
    constexpr int ROWS_PER_PIECE = NT / SPR;                                // LDS rows per workgroup-wide piece
    __shared__ __attribute__((aligned(1024))) char smem[2 * STAGE];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 31, lk = lane >> 5;
    const int qblocks = p.T / (QW * 32);
    // Workgroups are dealt to the 8 XCDs round-robin by blockIdx.  The T / 128 query blocks of one (sample, head) all stream that head's
    // whole K and V^T: in blockIdx order they land on DIFFERENT XCDs and every L2 fetches every head (PMC, round 6: 418 MB fetched per
    // launch at 32 x 32 against 201 MB of qkv).  XCD x owns a contiguous range of logical blocks instead (the convolution kernels' map).
    int bid = blockIdx.x;
    if (p.xcd_map) {
        const int x = bid & 7, per = gridDim.x >> 3, rem = gridDim.x & 7;
        bid = (x < rem ? x * (per + 1) : rem * (per + 1) + (x - rem) * per) + (bid >> 3);
    }
    const int z = bid / qblocks, qb = bid - z * qblocks;
    const int q0 = qb * (QW * 32) + wave * 32;
    const int nkb = p.T / KB;

    // ---- staging: the stage image is STAGE/128 rows of 128 B: rows [0, SL*32) = K (slice-major), then D rows of V^T
    const int r_in_piece = tid / SPR, ps = tid % SPR;
    const int zb = z / p.NH, zh = z - zb * p.NH;
    static_assert((SL * KB) % ROWS_PER_PIECE == 0, "a DMA piece is all K rows or all V^T rows");
    const char* src[PIECES];
    // bytes to advance the source per key block: piece i is K rows for i < SL*KB / ROWS_PER_PIECE, V^T rows after (compile time)
#pragma unroll
    for (int i = 0; i < PIECES; ++i) {
        const int row = i * ROWS_PER_PIECE + r_in_piece;
        const int ls = H1 ? ps ^ ((row >> 2) & 3) : ps ^ ((row >> 1) & 7);
        if (row < SL * KB) {
            const int sl = row / KB, key = row - sl * KB;
            if constexpr (H1) src[i] = p.kh + ((size_t)zb * p.T + key) * p.row_bytes + (size_t)zh * p.head_bytes + sl * 64 + ls * 16;
            else src[i] = p.kh + (((size_t)z * p.T + key) * D) * 4 + sl * 128 + ls * 16;
        } else {
            const int dd = row - SL * KB;
            src[i] = p.vt + (((size_t)z * nkb) * D + dd) * ROWB + ls * 16;
        }
    }
    const int wdst = wave * 8 * 128;        // this wave's first row inside a piece (wave-uniform)
    auto issue = [&](int stage) {
        char* base = smem + stage * STAGE + wdst;
#pragma unroll
        for (int i = 0; i < PIECES; ++i) {
            AT_GLDS(src[i], base + i * ROWS_PER_PIECE * ROWB);
            src[i] += (i * ROWS_PER_PIECE < SL * KB) ? (H1 ? (size_t)KB * p.row_bytes : (size_t)KB * D * 4) : (size_t)D * ROWB;
        }
    };

    // ---- Q fragments (B operand: column = query lr, k = 8 consecutive d of octet ks*2 + lk), kept in registers
    half8 qh[D / 16], ql[H1 ? 1 : D / 16];
    if constexpr (H1) {
        const char* qrow = p.qh + ((size_t)zb * p.T + q0 + lr) * p.row_bytes + (size_t)zh * p.head_bytes;
#pragma unroll
        for (int ks = 0; ks < D / 16; ++ks) qh[ks] = *reinterpret_cast<const half8*>(qrow + (ks * 2 + lk) * 16);
    } else {
        const char* qrow = p.qh + (((size_t)z * p.T + q0 + lr) * D) * 4;
#pragma unroll
        for (int ks = 0; ks < D / 16; ++ks) {
            const half8* s = reinterpret_cast<const half8*>(qrow + (ks * 2 + lk) * 32);
            qh[ks] = s[0];
            ql[ks] = s[1];
        }
    }

    f32x16 o[D / 32];
#pragma unroll
    for (int t = 0; t < D / 32; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[t][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;

    issue(0);
    for (int kb = 0; kb < nkb; ++kb) {
        const int cur = kb & 1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (kb + 1 < nkb) issue(cur ^ 1);
        const char* Ks = smem + cur * STAGE;
        const char* Vs = Ks + KT;

        // S^T = K_blk Q^T
        f32x16 s;
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < D / 16; ++ks) {
            const int sl = ks >> 1, st = ks & 1;
            const char* rowp = Ks + sl * (KB * ROWB);
            if constexpr (H1) {
                const half8 kh = *reinterpret_cast<const half8*>(rowp + swz64(lr, st * 2 + lk));
                s = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, qh[ks], s, 0, 0, 0);
            } else {
                const half8 kh = *reinterpret_cast<const half8*>(rowp + swz128(lr, st * 4 + lk * 2));
                const half8 kl = *reinterpret_cast<const half8*>(rowp + swz128(lr, st * 4 + lk * 2 + 1));
                s = __builtin_amdgcn_mfma_f32_32x32x16_f16(kl, qh[ks], s, 0, 0, 0);
                s = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, ql[ks], s, 0, 0, 0);
                s = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, qh[ks], s, 0, 0, 0);
            }
        }
        if constexpr (H1) {                 // 1 / sqrt(d): the three-pass form has it on the packed Q
#pragma unroll
            for (int r = 0; r < 16; ++r) s[r] *= p.qscale;
        }

        // online softmax for query lr: this lane holds 16 of the block's 32 keys, lane ^ 32 the other 16
        float mb = s[0];
#pragma unroll
        for (int r = 1; r < 16; ++r) mb = fmaxf(mb, s[r]);
        mb = fmaxf(mb, __shfl_xor(mb, 32, 64));
        const float m_new = fmaxf(m_run, mb);
        const float alpha = __expf(m_run - m_new);
        float psum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            s[r] = __expf(s[r] - m_new);
            psum += s[r];
        }
        psum += __shfl_xor(psum, 32, 64);
        l_run = l_run * alpha + psum;
        m_run = m_new;
#pragma unroll
        for (int t = 0; t < D / 32; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[t][r] *= alpha;

        // O^T += V^T_blk P^T : registers r = 8*g .. 8*g+7 are the B operand of key group g as they stand
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            half8 ph, pl;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float v = s[g * 8 + j];
                ph[j] = (_Float16)v;
                pl[j] = (_Float16)(v - (float)ph[j]);
            }
#pragma unroll
            for (int t = 0; t < D / 32; ++t) {
                if constexpr (H1) {
                    const half8 vh = *reinterpret_cast<const half8*>(Vs + swz64(t * 32 + lr, g * 2 + lk));
                    o[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, ph, o[t], 0, 0, 0);
                } else {
                    const half8 vh = *reinterpret_cast<const half8*>(Vs + swz128(t * 32 + lr, g * 4 + lk * 2));
                    const half8 vl = *reinterpret_cast<const half8*>(Vs + swz128(t * 32 + lr, g * 4 + lk * 2 + 1));
                    o[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vl, ph, o[t], 0, 0, 0);
                    o[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, pl, o[t], 0, 0, 0);
                    o[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, ph, o[t], 0, 0, 0);
                }
            }
        }
    }

    // ---- O^T tile t: rows = d (t*32 + (r&3) + 8*(r>>2) + 4*lk), column = query lr -> out[b][q0 + lr][h*D + d]
    const int b = z / p.NH, h = z - b * p.NH;
    const float inv = 1.f / l_run;
    if (p.out16) {      // token (ty, tx) -> interior pixel (ty + 1, tx + 1) of the bordered fp16 operand of proj_out
        typedef _Float16 half4 __attribute__((ext_vector_type(4)));
        const int tok = q0 + lr, ty = tok / p.W, tx = tok - ty * p.W;
        _Float16* orow = reinterpret_cast<_Float16*>(p.out) + (((size_t)b * (p.T / p.W + 2) + ty + 1) * (p.W + 2) + tx + 1) * p.C + h * D;
#pragma unroll
        for (int t = 0; t < D / 32; ++t)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const half4 v = {(_Float16)(o[t][g * 4] * inv), (_Float16)(o[t][g * 4 + 1] * inv), (_Float16)(o[t][g * 4 + 2] * inv),
                                 (_Float16)(o[t][g * 4 + 3] * inv)};
                *reinterpret_cast<half4*>(orow + t * 32 + g * 8 + lk * 4) = v;
            }
        return;
    }
    float* orow = p.out + ((size_t)b * p.T + q0 + lr) * p.C + h * D;
#pragma unroll
    for (int t = 0; t < D / 32; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            f32x4 v = {o[t][g * 4] * inv, o[t][g * 4 + 1] * inv, o[t][g * 4 + 2] * inv, o[t][g * 4 + 3] * inv};
            *reinterpret_cast<f32x4*>(orow + t * 32 + g * 8 + lk * 4) = v;
        }
}

}  // namespace

extern "C" int dp_attention_fused(const void* qkv, int qkv_fmt, int B, int T, int C, int n_heads, int layout, void* out, int out_fmt, int W,
                                  void* work, void* stream) {
    DP_REQUIRE(qkv && out && work && B > 0 && T > 0 && C > 0 && n_heads > 0, "dp_attention_fused: bad args");
    DP_REQUIRE(qkv_fmt == 0 || qkv_fmt == 1, "dp_attention_fused: qkv_fmt %d (0 = fp32, three-pass arithmetic; 1 = fp16, one pass)", qkv_fmt);
    const int D = C / n_heads;
    DP_REQUIRE(C % n_heads == 0 && (D == 64 || D == 256), "dp_attention_fused: head dimension must be 64 or 256 (got %d)", D);
    DP_REQUIRE(T % 64 == 0 && (D == 64 || T % 128 == 0), "dp_attention_fused: token count must be a multiple of 64 (128 at head dimension 256), got %d", T);
    DP_REQUIRE(layout == 0 || layout == 1, "dp_attention_fused: layout %d", layout);
    DP_REQUIRE(out_fmt == 0 || (out_fmt == 1 && W > 0 && T % W == 0), "dp_attention_fused: out_fmt 1 (bordered fp16 operand) needs the image width W | T (got %d, W=%d)", out_fmt, W);
    DP_REQUIRE(dp_aligned16(qkv) && dp_aligned16(out) && dp_aligned16(work), "dp_attention_fused: misaligned tensor");
    hipStream_t s = (hipStream_t)stream;
    const int Z = B * n_heads;
    int oq, ok, ov, sh;
    if (layout == 0) { oq = 0; ok = D; ov = 2 * D; sh = 3 * D; }       // 'legacy': heads x [q | k | v]
    else { oq = 0; ok = C; ov = 2 * C; sh = D; }                       // 'split' : [Q | K | V]
    const float qscale = 1.0f / sqrtf((float)D);
    if (qkv_fmt == 1) {                     // one fp16 pass: Q and K are read in place, only V^T is packed
        const _Float16* q16 = static_cast<const _Float16*>(qkv);
        PackVtArgs pv{q16, T, C, n_heads, ov, sh, (char*)work};
        if (D == 64) hipLaunchKernelGGL(attn_pack_vt_kernel<64>, dim3((unsigned)(Z * (T / KB))), dim3(256), 0, s, pv);
        else hipLaunchKernelGGL(attn_pack_vt_kernel<256>, dim3((unsigned)(Z * (T / KB))), dim3(256), 0, s, pv);
        DP_LAUNCH_CHECK("attn_pack_vt");
        FlashArgs fa{reinterpret_cast<const char*>(q16 + oq), reinterpret_cast<const char*>(q16 + ok), (const char*)work, static_cast<float*>(out),
                     T, C, n_heads, out_fmt, out_fmt ? W : 1, 3 * C * 2, sh * 2, qscale, dp_tune(DP_T_XCD_MAP) != 0};
        if (D == 256) hipLaunchKernelGGL((attn_flash_kernel<4, 256, true>), dim3((unsigned)(Z * (T / 128))), dim3(256), 0, s, fa);
        else if (T % 128 == 0) hipLaunchKernelGGL((attn_flash_kernel<4, 64, true>), dim3((unsigned)(Z * (T / 128))), dim3(256), 0, s, fa);
        else hipLaunchKernelGGL((attn_flash_kernel<2, 64, true>), dim3((unsigned)(Z * (T / 64))), dim3(128), 0, s, fa);
        DP_LAUNCH_CHECK("attn_flash");
        return 0;
    }
    const size_t part = (size_t)B * T * C * 4;
    PackArgs pa;
    pa.qkv = static_cast<const float*>(qkv); pa.B = B; pa.T = T; pa.C = C; pa.NH = n_heads;
    pa.oq = oq; pa.ok = ok; pa.ov = ov; pa.sh = sh;
    pa.qscale = qscale;
    pa.qh = (char*)work; pa.kh = pa.qh + part; pa.vt = pa.kh + part;
    if (D == 64) hipLaunchKernelGGL(attn_pack_kernel<64>, dim3((unsigned)(Z * (T / KB))), dim3(256), 0, s, pa);
    else hipLaunchKernelGGL(attn_pack_kernel<256>, dim3((unsigned)(Z * (T / KB))), dim3(256), 0, s, pa);
    DP_LAUNCH_CHECK("attn_pack");
    FlashArgs fa{pa.qh, pa.kh, pa.vt, static_cast<float*>(out), T, C, n_heads, out_fmt, out_fmt ? W : 1, 0, 0, 1.0f, dp_tune(DP_T_XCD_MAP) != 0};
    if (D == 256) hipLaunchKernelGGL((attn_flash_kernel<4, 256, false>), dim3((unsigned)(Z * (T / 128))), dim3(256), 0, s, fa);
    else if (T % 128 == 0) hipLaunchKernelGGL((attn_flash_kernel<4, 64, false>), dim3((unsigned)(Z * (T / 128))), dim3(256), 0, s, fa);
    else hipLaunchKernelGGL((attn_flash_kernel<2, 64, false>), dim3((unsigned)(Z * (T / 64))), dim3(128), 0, s, fa);
    DP_LAUNCH_CHECK("attn_flash");
    return 0;
}
