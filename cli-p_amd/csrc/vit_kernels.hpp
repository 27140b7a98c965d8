// vit_kernels.hpp — the non-GEMM kernels of the CLIP towers (gfx950): patch extraction with the
// fused pixel normalisation, CLS/positional rows, LayerNorm, fused small-sequence attention,
// token embedding and EOT pooling. Reference call sites: model.encode_image (build-index.py:49),
// model.encode_text (query-index.py:108); op list in SURVEY.md §2.1.
#pragma once
#include "gemm.hpp"

namespace clipmi {

// ---------------------------------------------------------------------------------------------
// patches: pixels [B][3][R][R] (f32 / bf16 normalised, or u8 raw) -> bf16 [B*np][patch_k],
// column k = c*P*P + py*P + px (conv1.weight flattened), zero beyond 3*P*P. One thread = 8 px.
// u8 input fuses CLIP's transform tail (x/255 - mean)/std.
// ---------------------------------------------------------------------------------------------
struct PatchArgs {
    const void* pix;
    unsigned short* out;
    int dtype, B, R, P, grid, np, patch_k;
};

static __global__ void __launch_bounds__(256) patchify_kernel(PatchArgs a) {
    const long long total = (long long)a.B * a.np * (a.patch_k / 8);
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int kc = a.patch_k / 8;
    const int k8 = (int)(i % kc);
    const long long bp = i / kc;
    const int p = (int)(bp % a.np);
    const int b = (int)(bp / a.np);
    const int PP = a.P * a.P;
    const float mean[3] = {0.48145466f, 0.4578275f, 0.40821073f};
    const float stdv[3] = {0.26862954f, 0.26130258f, 0.27577711f};
    float v[8];
    if ((a.P & 7) == 0 && (a.R & 7) == 0) {
        // 8 consecutive pixels of one patch row: one aligned vector load (8 B for u8, 16 B for bf16,
        // 2 x 16 B for f32) instead of 8 scalar ones
        const int k = k8 * 8;
        if (k < 3 * PP) {
            const int c = k / PP, rem = k - c * PP;
            const int py = rem / a.P, px = rem - py * a.P;
            const int y = (p / a.grid) * a.P + py, xx = (p % a.grid) * a.P + px;
            const size_t src = (((size_t)b * 3 + c) * a.R + y) * a.R + xx;
            if (a.dtype == CLIPMI_F32) {
                const f32x4 lo = *reinterpret_cast<const f32x4*>(static_cast<const float*>(a.pix) + src);
                const f32x4 hi = *reinterpret_cast<const f32x4*>(static_cast<const float*>(a.pix) + src + 4);
                v[0] = lo.x; v[1] = lo.y; v[2] = lo.z; v[3] = lo.w; v[4] = hi.x; v[5] = hi.y; v[6] = hi.z; v[7] = hi.w;
            } else if (a.dtype == CLIPMI_BF16) {
                const uint4 raw = *reinterpret_cast<const uint4*>(static_cast<const unsigned short*>(a.pix) + src);
                const unsigned w[4] = {raw.x, raw.y, raw.z, raw.w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    v[2 * j] = __uint_as_float(w[j] << 16);
                    v[2 * j + 1] = __uint_as_float(w[j] & 0xffff0000u);
                }
            } else {
                const uint2 raw = *reinterpret_cast<const uint2*>(static_cast<const unsigned char*>(a.pix) + src);
                const float m = mean[c], s = stdv[c];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const unsigned byte = ((j < 4 ? raw.x : raw.y) >> (8 * (j & 3))) & 0xffu;
                    v[j] = ((float)byte / 255.0f - m) / s;
                }
            }
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = 0.f;
        }
    } else {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int k = k8 * 8 + j;
        float x = 0.f;
        if (k < 3 * PP) {
            const int c = k / PP, rem = k - c * PP;
            const int py = rem / a.P, px = rem - py * a.P;
            const int y = (p / a.grid) * a.P + py, xx = (p % a.grid) * a.P + px;
            const size_t src = (((size_t)b * 3 + c) * a.R + y) * a.R + xx;
            if (a.dtype == CLIPMI_F32) x = static_cast<const float*>(a.pix)[src];
            else if (a.dtype == CLIPMI_BF16) x = bf16_to_f32(static_cast<const unsigned short*>(a.pix)[src]);
            else x = ((float)static_cast<const unsigned char*>(a.pix)[src] / 255.0f - mean[c]) / stdv[c];
        }
        v[j] = x;
    }
    }
    uint4 o = make_uint4(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7]));
    *reinterpret_cast<uint4*>(a.out + ((size_t)bp * a.patch_k + k8 * 8)) = o;
}

// u8 fast path of patchify_kernel for P % 16 == 0, R % 16 == 0 and patch_k == 3 P^2 (ViT-B/32, ViT-B/16): one workgroup
// per (image, channel, patch row) stages that strip - P image rows = P*R CONTIGUOUS bytes, whole lines - in LDS and writes
// the strip's patches in OUTPUT order, 16 bytes per lane, 1 KiB contiguous per wave-instruction. patchify_kernel maps
// threads along the output without staging: a wave then reads sixteen 32-byte pieces from sixteen lines whose other 96
// bytes belong to other workgroups (2.8 TB/s of useful bytes at B = 870); mapping threads along the input instead makes
// the stores the scattered side (3.5 TB/s).
static __global__ void __launch_bounds__(256) patchify_strip_u8_kernel(PatchArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int gy = blockIdx.x % a.grid;
    const int c = (blockIdx.x / a.grid) % 3;
    const int b = blockIdx.x / (3 * a.grid);
    const float mean[3] = {0.48145466f, 0.4578275f, 0.40821073f};
    const float stdv[3] = {0.26862954f, 0.26130258f, 0.27577711f};
    const float m = mean[c], sd = stdv[c];
    const unsigned char* src = static_cast<const unsigned char*>(a.pix) + (((size_t)b * 3 + c) * a.R + (size_t)gy * a.P) * a.R;
    // the value of a pixel depends on (byte, channel) only: 256 results of patchify_kernel's expression per workgroup, then
    // one 2-byte LDS lookup per pixel (the two f32 divisions per pixel made the pass VALU-bound: ~30 instructions each)
    unsigned short* lut = reinterpret_cast<unsigned short*>(smem + a.P * a.R);
    lut[threadIdx.x] = (unsigned short)(pack_bf16x2(((float)threadIdx.x / 255.0f - m) / sd, 0.f) & 0xffffu);
    const int nchunk = (a.P * a.R) >> 4;
    for (int j = threadIdx.x; j < nchunk; j += 256)
        *reinterpret_cast<uint4*>(smem + j * 16) = *reinterpret_cast<const uint4*>(src + (size_t)j * 16);
    __syncthreads();
    const int per_patch = (a.P * a.P) >> 3, per_row = a.P >> 3;          // 8-pixel output pieces
    for (int o = threadIdx.x; o < a.grid * per_patch; o += 256) {
        const int gx = o / per_patch, rem = o - gx * per_patch;
        const int y = rem / per_row, px = (rem - y * per_row) << 3;
        const uint2 raw = *reinterpret_cast<const uint2*>(smem + y * a.R + gx * a.P + px);
        unsigned w4[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const unsigned word = k < 2 ? raw.x : raw.y;
            const unsigned b0 = (word >> (16 * (k & 1))) & 0xffu, b1 = (word >> (16 * (k & 1) + 8)) & 0xffu;
            w4[k] = (unsigned)lut[b0] | ((unsigned)lut[b1] << 16);
        }
        unsigned short* dst = a.out + ((size_t)b * a.np + gy * a.grid + gx) * a.patch_k + c * a.P * a.P + y * a.P + px;
        *reinterpret_cast<uint4*>(dst) = make_uint4(w4[0], w4[1], w4[2], w4[3]);
    }
}

// x[b*L + 0][:] = class_embedding + positional_embedding[0]
static __global__ void __launch_bounds__(256) cls_rows_kernel(float* x, const float* cls, const float* pos, int B, int L, int W) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)B * W) return;
    const int b = (int)(i / W), c = (int)(i - (long long)b * W);
    x[(size_t)b * L * W + c] = cls[c] + pos[c];
}

// x[q*L + t][:] = token_embedding[ids[q][t]] + positional_embedding[t]; one thread = 4 floats
static __global__ void __launch_bounds__(256) text_embed_kernel(float* x, const int* ids, const float* tok, const float* pos,
                                                         int Q, int L, int W, int vocab) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    const int wc = W / 4;
    if (i >= (long long)Q * L * wc) return;
    const int c4 = (int)(i % wc);
    const long long row = i / wc;
    const int t = (int)(row % L);
    int id = ids[row];
    id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
    const f32x4 e = *reinterpret_cast<const f32x4*>(tok + (size_t)id * W + c4 * 4);
    const f32x4 p = *reinterpret_cast<const f32x4*>(pos + (size_t)t * W + c4 * 4);
    *reinterpret_cast<f32x4*>(x + (size_t)row * W + c4 * 4) = e + p;
}

// rowidx[q] = q*L + (first argmax of ids[q][:])  — the EOT row of each prompt
static __global__ void eot_rows_kernel(const int* ids, int* rowidx, int Q, int L) {
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= Q) return;
    int best = ids[(size_t)q * L], bi = 0;
    for (int t = 1; t < L; ++t) {
        const int v = ids[(size_t)q * L + t];
        if (v > best) { best = v; bi = t; }
    }
    rowidx[q] = q * L + bi;
}

// ---------------------------------------------------------------------------------------------
// LayerNorm(eps 1e-5) over rows of f32 x: one wave per row, values held in registers,
// two-pass mean / variance as torch does. Source row r is x[rowidx ? rowidx[r] : r*row_step].
// ---------------------------------------------------------------------------------------------
struct LnArgs {
    const float* x;
    const float* w;
    const float* b;
    void* out;            // [M][W] bf16 or f32 (dense rows)
    const int* rowidx;    // optional gather
    long long row_step;   // source row stride in ROWS when rowidx == nullptr (1 = dense, L = CLS rows)
    int M, W, out_bf16;
    // FP8 path: when out8 is set, the row goes out as OCP e4m3 + one f32 scale instead of bf16 - exactly what
    // quantize_rows_fp8_kernel would make of the bf16 row (the values are rounded to bf16 first)
    unsigned char* out8;
    float* scale8;
    // split-residual input (gemm.hpp): when x == nullptr the source rows are the split rows x3 ([W bf16 hi | W u8 lo] each)
    const void* x3;
    // split-residual OUTPUT (ln_pre of an LN-folded tower): when out_x3 is set the normalised row leaves as a split row
    // plus its canonical statistics partials [M][W/256][2] (what split_stats_kernel would make of it)
    void* out_x3;
    float* out_part;
};

static __global__ void __launch_bounds__(256) layernorm_kernel(LnArgs a) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= a.M) return;
    const long long src_row = a.rowidx ? (long long)a.rowidx[r] : (long long)r * a.row_step;
    const float* p = a.x + src_row * a.W;
    f32x4 v[4];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = i * 256 + lane * 4;
        v[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (c < a.W) {
            if (a.x) v[i] = *reinterpret_cast<const f32x4*>(p + c);
            else v[i] = split_join(*reinterpret_cast<const uint2*>(resid_hi(a.x3, (size_t)src_row, a.W) + c),
                                   *reinterpret_cast<const unsigned*>(resid_lo(a.x3, (size_t)src_row, a.W) + c));
        }
        s += v[i].x + v[i].y + v[i].z + v[i].w;
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o);
    const float mean = s / (float)a.W;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = i * 256 + lane * 4;
        if (c < a.W) {
            const f32x4 d = v[i] - mean;
            q += d.x * d.x + d.y * d.y + d.z * d.z + d.w * d.w;
        }
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) q += __shfl_xor(q, o);
    const float rstd = rsqrtf(q / (float)a.W + 1e-5f);
    if (a.out8) {
        float mx = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = i * 256 + lane * 4;
            if (c < a.W) {
                const f32x4 g = *reinterpret_cast<const f32x4*>(a.w + c);
                const f32x4 bb = *reinterpret_cast<const f32x4*>(a.b + c);
                const f32x4 y = (v[i] - mean) * rstd * g + bb;
                const unsigned lo = pack_bf16x2(y.x, y.y), hi = pack_bf16x2(y.z, y.w);     // the bf16 row, kept in v[i]
                v[i] = f32x4{__uint_as_float(lo << 16), __uint_as_float(lo & 0xffff0000u), __uint_as_float(hi << 16),
                             __uint_as_float(hi & 0xffff0000u)};
                mx = fmaxf(fmaxf(mx, fabsf(v[i].x)), fmaxf(fabsf(v[i].y), fmaxf(fabsf(v[i].z), fabsf(v[i].w))));
            }
        }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
        const float sc = mx > 0.f ? mx / 448.0f : 1.0f;
        const float inv = 1.0f / sc;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = i * 256 + lane * 4;
            if (c < a.W) {
                int pk = 0;
                pk = __builtin_amdgcn_cvt_pk_fp8_f32(v[i].x * inv, v[i].y * inv, pk, false);
                pk = __builtin_amdgcn_cvt_pk_fp8_f32(v[i].z * inv, v[i].w * inv, pk, true);
                *reinterpret_cast<unsigned*>(a.out8 + (size_t)r * a.W + c) = (unsigned)pk;
            }
        }
        if (lane == 0) a.scale8[r] = sc;
        return;
    }
    if (a.out_x3) {              // W % 256 == 0 (checked by the launcher): segment i = columns 256 i .. 256 i + 255
        const int nseg = a.W >> 8;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (i < nseg) {
                const int c = i * 256 + lane * 4;
                const f32x4 g = *reinterpret_cast<const f32x4*>(a.w + c);
                const f32x4 bb = *reinterpret_cast<const f32x4*>(a.b + c);
                const f32x4 y = (v[i] - mean) * rstd * g + bb;
                uint2 nh;
                unsigned nl;
                split_make(y, nh, nl);
                *reinterpret_cast<uint2*>(resid_hi(a.out_x3, (size_t)r, a.W) + c) = nh;
                *reinterpret_cast<unsigned*>(resid_lo(a.out_x3, (size_t)r, a.W) + c) = nl;
                const float sa = ln_wave_sum(ln_lane_sum(y));
                const float sq = ln_wave_sum(ln_lane_sumsq(y));
                if (lane == 0) *reinterpret_cast<f32x2*>(a.out_part + ((size_t)r * nseg + i) * 2) = f32x2{sa, sq};
            }
        }
        return;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = i * 256 + lane * 4;
        if (c < a.W) {
            const f32x4 g = *reinterpret_cast<const f32x4*>(a.w + c);
            const f32x4 bb = *reinterpret_cast<const f32x4*>(a.b + c);
            const f32x4 y = (v[i] - mean) * rstd * g + bb;
            if (a.out_bf16)
                *reinterpret_cast<uint2*>(static_cast<unsigned short*>(a.out) + (size_t)r * a.W + c) =
                    make_uint2(pack_bf16x2(y.x, y.y), pack_bf16x2(y.z, y.w));
            else
                *reinterpret_cast<f32x4*>(static_cast<float*>(a.out) + (size_t)r * a.W + c) = y;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// LN-folded layers (gemm.hpp): the stand-alone producer of the split residual and its row statistics,
//   rows = ADD ? add[m][:] + x3[m][:] : x[m][:];   x3[m] = split_make(rows) (gemm.hpp: W bf16 hi | W u8 lo),
//   part[m][j] = (sum, sum of squares) of columns 256 j .. 256 j + 255
// used where the persistent residual GEMM's own store pass is not available (the embedded rows after ln_pre; residual
// GEMMs that run on the non-persistent kernels, whose acc + bias arrive as f32 in `add`). One wave per row,
// W % 256 == 0, W <= 1024; statistics in the canonical order (same bits as gemm256p's storers).
// ---------------------------------------------------------------------------------------------
template <bool ADD>
static __global__ void __launch_bounds__(256) split_stats_kernel(const float* __restrict__ x, void* x3, float* __restrict__ part,
                                                                 int M, int W) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= M) return;
    const int nseg = W >> 8;
    const float* p = x + (size_t)r * W;
    unsigned short* const hrow = resid_hi(x3, (size_t)r, W);
    unsigned char* const lrow = resid_lo(x3, (size_t)r, W);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if (j < nseg) {
            const int c = j * 256 + lane * 4;
            f32x4 v = *reinterpret_cast<const f32x4*>(p + c);
            if (ADD) v = v + split_join(*reinterpret_cast<const uint2*>(hrow + c), *reinterpret_cast<const unsigned*>(lrow + c));
            uint2 nh;
            unsigned nl;
            split_make(v, nh, nl);
            *reinterpret_cast<uint2*>(hrow + c) = nh;
            *reinterpret_cast<unsigned*>(lrow + c) = nl;
            const float sa = ln_wave_sum(ln_lane_sum(v));
            const float sq = ln_wave_sum(ln_lane_sumsq(v));
            if (lane == 0) *reinterpret_cast<f32x2*>(part + ((size_t)r * nseg + j) * 2) = f32x2{sa, sq};
        }
    }
}

// text_embed_kernel + split_stats_kernel<false> + eot_rows_kernel in ONE launch (LN-folded text towers; round 5: one prompt is a
// chain of ~65 dependent launches of ~4 us, the three head kernels were three of them). Blocks [0, ceil(M / 4)): one wave per
// row - token + positional embedding, split_make, canonical statistics, exactly the bits of the three-kernel path; block
// ceil(M / 4): the EOT rows.
static __global__ void __launch_bounds__(256) text_embed_split_kernel(const int* __restrict__ ids, const float* __restrict__ tok,
                                                                      const float* __restrict__ pos, void* x3, float* __restrict__ part,
                                                                      int* __restrict__ rowidx, int Q, int L, int W, int vocab) {
    const int M = Q * L, nrb = (M + 3) / 4;
    if ((int)blockIdx.x >= nrb) {
        for (int q = threadIdx.x; q < Q; q += 256) {
            int best = ids[(size_t)q * L], bi = 0;
            for (int t = 1; t < L; ++t) {
                const int v = ids[(size_t)q * L + t];
                if (v > best) { best = v; bi = t; }
            }
            rowidx[q] = q * L + bi;
        }
        return;
    }
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= M) return;
    const int nseg = W >> 8, t = r % L;
    int id = ids[r];
    id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
    unsigned short* const hrow = resid_hi(x3, (size_t)r, W);
    unsigned char* const lrow = resid_lo(x3, (size_t)r, W);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if (j < nseg) {
            const int c = j * 256 + lane * 4;
            const f32x4 v = *reinterpret_cast<const f32x4*>(tok + (size_t)id * W + c) + *reinterpret_cast<const f32x4*>(pos + (size_t)t * W + c);
            uint2 nh;
            unsigned nl;
            split_make(v, nh, nl);
            *reinterpret_cast<uint2*>(hrow + c) = nh;
            *reinterpret_cast<unsigned*>(lrow + c) = nl;
            const float sa = ln_wave_sum(ln_lane_sum(v));
            const float sq = ln_wave_sum(ln_lane_sumsq(v));
            if (lane == 0) *reinterpret_cast<f32x2*>(part + ((size_t)r * nseg + j) * 2) = f32x2{sa, sq};
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Fused attention for short sequences (L <= 16*NT; ViT-B/32: L = 50, text: L = 77), head dim 64.
// One wave per (sequence, head). qkv bf16 [B*L][3W] (q | k | v, head h at columns 64h..64h+63
// of each third) -> out bf16 [B*L][W].
//
// S^T = K Q^T is computed with the KEY on the accumulator rows and the QUERY on its column
// (lane & 15): the softmax reduction over keys is then 4*NT in-lane values plus two xor-shuffles,
// and the P^T accumulator registers ARE the B operand of O^T = V^T P^T with no data movement
// (k-slot j of lane group g is key 16*(2s + (j>>2)) + 4g + (j&3), for both operands).
// K and Q fragments come straight from global memory (16 B per lane); V goes through a per-wave
// LDS tile and is read transposed with ds_read_b64_tr_b16 (TR = 1) or 2-byte reads (TR = 0).
// Scores, softmax and normalisation are f32; P and the output are rounded to bf16.
// ---------------------------------------------------------------------------------------------
typedef short s16x4 __attribute__((ext_vector_type(4)));

template <int NT, bool CAUSAL, bool TR>
__global__ void __launch_bounds__(256) attention_kernel(const unsigned short* __restrict__ qkv,
                                                        unsigned short* __restrict__ out, int B, int L, int heads) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int KS = (NT + 1) / 2;            // 32-key steps of the second product
    constexpr int ROWS = KS * 32;               // V rows staged (>= 16*NT), zero-weight beyond L
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int fr = lane & 15, fg = lane >> 4;
    const int W = heads * 64;
    const long long item = (long long)blockIdx.x * 4 + wave;      // (b, h)
    if (item >= (long long)B * heads) return;                       // wave-uniform; no block barriers below
    const int b = (int)(item / heads), h = (int)(item - (long long)b * heads);
    const unsigned short* base = qkv + (size_t)b * L * 3 * W + h * 64;
    const size_t rs = (size_t)3 * W;                                // row stride (elements)
    char* vt = smem + wave * (ROWS * 128);

    // ---- stage V rows [0, ROWS) of this head: 8 rows x 128 B per wave-instruction
#pragma unroll
    for (int i = 0; i < ROWS / 8; ++i) {
        int row = i * 8 + (lane >> 3);
        const int srcrow = row < L ? row : L - 1;
        const uint4 d = *reinterpret_cast<const uint4*>(base + (size_t)srcrow * rs + 2 * W + (lane & 7) * 8);
        *reinterpret_cast<uint4*>(vt + row * 128 + (lane & 7) * 16) = d;
    }

    // ---- S^T tiles: acc[kt][qt], rows = keys 16kt + 4fg + r, column = query 16qt + fr
    bf16x8 kf[NT][2], qf[NT][2];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        int row = t * 16 + fr;
        row = row < L ? row : L - 1;
        const unsigned short* pr = base + (size_t)row * rs + fg * 8;
        qf[t][0] = *reinterpret_cast<const bf16x8*>(pr);
        qf[t][1] = *reinterpret_cast<const bf16x8*>(pr + 32);
        kf[t][0] = *reinterpret_cast<const bf16x8*>(pr + W);
        kf[t][1] = *reinterpret_cast<const bf16x8*>(pr + W + 32);
    }
    f32x4 s[NT][NT];
#pragma unroll
    for (int kt = 0; kt < NT; ++kt)
#pragma unroll
        for (int qt = 0; qt < NT; ++qt) {
            f32x4 a = {0.f, 0.f, 0.f, 0.f};
            a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[kt][0], qf[qt][0], a, 0, 0, 0);
            a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[kt][1], qf[qt][1], a, 0, 0, 0);
            s[kt][qt] = a;
        }

    // ---- softmax over keys for each query column (f32): scale 1/8, mask, max, exp, sum
#pragma unroll
    for (int qt = 0; qt < NT; ++qt) {
        const int qi = qt * 16 + fr;
        float mx = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < NT; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int ki = kt * 16 + 4 * fg + r;
                float x = s[kt][qt][r] * 0.125f;
                if (ki >= L || (CAUSAL && ki > qi)) x = -INFINITY;
                s[kt][qt][r] = x;
                mx = fmaxf(mx, x);
            }
        mx = fmaxf(mx, __shfl_xor(mx, 16));
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        float sum = 0.f;
#pragma unroll
        for (int kt = 0; kt < NT; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float e = __expf(s[kt][qt][r] - mx);     // key 0 is never masked: mx is finite
                s[kt][qt][r] = e;
                sum += e;
            }
        sum += __shfl_xor(sum, 16);
        sum += __shfl_xor(sum, 32);
        const float inv = 1.0f / sum;
#pragma unroll
        for (int kt = 0; kt < NT; ++kt) s[kt][qt] *= inv;
    }

    // ---- O^T = V^T P^T: acc o[dt][qt], rows = d 16dt + 4fg + r, column = query
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // this wave's V tile is in LDS
    __builtin_amdgcn_wave_barrier();
    f32x4 o[4][NT];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
        for (int qt = 0; qt < NT; ++qt) o[dt][qt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        bf16x8 pf[NT];
#pragma unroll
        for (int qt = 0; qt < NT; ++qt) {
            const f32x4 lo = s[2 * ks][qt];
            f32x4 hi = {0.f, 0.f, 0.f, 0.f};
            if (2 * ks + 1 < NT) hi = s[2 * ks + 1 < NT ? 2 * ks + 1 : 0][qt];
            const uint4 u = make_uint4(pack_bf16x2(lo.x, lo.y), pack_bf16x2(lo.z, lo.w), pack_bf16x2(hi.x, hi.y),
                                       pack_bf16x2(hi.z, hi.w));
            pf[qt] = __builtin_bit_cast(bf16x8, u);
        }
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            bf16x8 vf;
            if (TR) {
                // 16-lane group fg reads the 4-key x 16-d block at keys 16*kt' + 4fg, d 16dt..16dt+15;
                // lane i of the group supplies the address of key (i>>2), d 4*(i&3) and receives d = i
                const int k0 = 32 * ks + 4 * fg + (fr >> 2);
                const char* ad = vt + k0 * 128 + (dt * 16 + 4 * (fr & 3)) * 2;
                const s16x4 t0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(ad));
                const s16x4 t1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(ad + 16 * 128));
                typedef short s16x8 __attribute__((ext_vector_type(8)));
                const s16x8 t = {t0.x, t0.y, t0.z, t0.w, t1.x, t1.y, t1.z, t1.w};
                vf = __builtin_bit_cast(bf16x8, t);
            } else {
                unsigned short e[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int key = 16 * (2 * ks + (j >> 2)) + 4 * fg + (j & 3);
                    e[j] = *reinterpret_cast<const unsigned short*>(vt + key * 128 + (dt * 16 + fr) * 2);
                }
                const uint4 u = make_uint4(e[0] | ((unsigned)e[1] << 16), e[2] | ((unsigned)e[3] << 16),
                                           e[4] | ((unsigned)e[5] << 16), e[6] | ((unsigned)e[7] << 16));
                vf = __builtin_bit_cast(bf16x8, u);
            }
#pragma unroll
            for (int qt = 0; qt < NT; ++qt)
                o[dt][qt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf[qt], o[dt][qt], 0, 0, 0);
        }
    }

    // ---- store: lane holds 4 consecutive d for query 16qt + fr
#pragma unroll
    for (int qt = 0; qt < NT; ++qt) {
        const int qi = qt * 16 + fr;
        if (qi >= L) continue;
        unsigned short* dst = out + ((size_t)b * L + qi) * W + h * 64 + 4 * fg;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            const f32x4 v = o[dt][qt];
            *reinterpret_cast<uint2*>(dst + dt * 16) = make_uint2(pack_bf16x2(v.x, v.y), pack_bf16x2(v.z, v.w));
        }
    }
}

// ---------------------------------------------------------------------------------------------
// quantize_rows_fp8_kernel: bf16 [M][K] -> OCP e4m3 [M][K] + f32 scale [M] (the A operand of gemm256f8).
// One wave per row: scale = max|x| / 448 (1 for an all-zero row), q = RNE(x * (1 / scale)) by v_cvt_pk_fp8_f32;
// dequantised value = scale * q. K % 8 == 0.
// ---------------------------------------------------------------------------------------------
static __global__ void __launch_bounds__(256) quantize_rows_fp8_kernel(const unsigned short* __restrict__ in,
                                                                       unsigned char* __restrict__ out,
                                                                       float* __restrict__ scale, int M, int K) {
    const int lane = threadIdx.x & 63;
    const long long r = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= M) return;
    const unsigned short* x = in + (size_t)r * K;
    float mx = 0.f;
    for (int k = lane * 8; k < K; k += 512) {
        const uint4 v = *reinterpret_cast<const uint4*>(x + k);
        const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int j = 0; j < 4; ++j)
            mx = fmaxf(mx, fmaxf(fabsf(__uint_as_float(w[j] << 16)), fabsf(__uint_as_float(w[j] & 0xffff0000u))));
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    const float s = mx > 0.f ? mx / 448.0f : 1.0f;
    const float inv = 1.0f / s;
    for (int k = lane * 8; k < K; k += 512) {
        const uint4 v = *reinterpret_cast<const uint4*>(x + k);
        const unsigned w[4] = {v.x, v.y, v.z, v.w};
        float f[8];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            f[2 * j] = __uint_as_float(w[j] << 16) * inv;
            f[2 * j + 1] = __uint_as_float(w[j] & 0xffff0000u) * inv;
        }
        int lo = 0, hi = 0;
        lo = __builtin_amdgcn_cvt_pk_fp8_f32(f[0], f[1], lo, false);
        lo = __builtin_amdgcn_cvt_pk_fp8_f32(f[2], f[3], lo, true);
        hi = __builtin_amdgcn_cvt_pk_fp8_f32(f[4], f[5], hi, false);
        hi = __builtin_amdgcn_cvt_pk_fp8_f32(f[6], f[7], hi, true);
        *reinterpret_cast<uint2*>(out + (size_t)r * K + k) = make_uint2((unsigned)lo, (unsigned)hi);
    }
    if (lane == 0) scale[r] = s;
}

// ---------------------------------------------------------------------------------------------
// MX block scales for e4m3 activations (gemm256f8.hpp BSA; round 3): 32 consecutive values of a row share the e8m0
// scale 2^(e - 7), e = floor(log2(largest |x| of the block)) - the scaled values then lie below 256 (e4m3 holds 448),
// a block of zeros takes 2^0. fp8mx_scale_byte gives the scale's e8m0 byte from the block's largest magnitude (a
// non-negative float: its exponent field - 7, clamped at 0), fp8mx_inv the exact reciprocal of the scale as a float.
// Every producer (the stand-alone pass below, the QuickGELU epilogue of gemm256p, attention52x4's output stage) uses
// these two, so the bytes do not depend on who produced them.
// ---------------------------------------------------------------------------------------------
// quantize_rows_fp8mx_kernel: bf16 [M][K] -> e4m3 [M][K] + e8m0 block scales [M][K / 32]; one wave per row, a lane
// takes 8 consecutive values per step, the four lanes of a quad make one block. K % 32 == 0. The fallback producer
// (shapes whose producing kernel has no fused form) and the reference the fused producers are tested against.
static __global__ void __launch_bounds__(256) quantize_rows_fp8mx_kernel(const unsigned short* __restrict__ in,
                                                                         unsigned char* __restrict__ out,
                                                                         unsigned char* __restrict__ bscale, int M, int K) {
    const int lane = threadIdx.x & 63;
    const long long r = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= M) return;
    const unsigned short* x = in + (size_t)r * K;
    for (int k0 = 0; k0 < K; k0 += 512) {
        const int k = k0 + lane * 8;
        const bool on = k < K;
        float f[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (on) {
            const uint4 v = *reinterpret_cast<const uint4*>(x + k);
            const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                f[2 * j] = __uint_as_float(w[j] << 16);
                f[2 * j + 1] = __uint_as_float(w[j] & 0xffff0000u);
            }
        }
        float mx = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) mx = fmaxf(mx, fabsf(f[j]));
        mx = fmaxf(mx, __shfl_xor(mx, 1));
        mx = fmaxf(mx, __shfl_xor(mx, 2));
        const unsigned sb = fp8mx_scale_byte(mx);
        const float inv = fp8mx_inv(sb);
#pragma unroll
        for (int j = 0; j < 8; ++j) f[j] *= inv;
        if (on) {
            *reinterpret_cast<uint2*>(out + (size_t)r * K + k) = fp8_pack8(f);
            if ((lane & 3) == 0) bscale[(size_t)r * (K >> 5) + (k >> 5)] = (unsigned char)sb;
        }
    }
}

// rows_mx_stats_kernel (FP8 towers with folded LayerNorms, round 4): f32 rows [M][W] -> the rows as e4m3 with MX block
// scales [M][W] + [M][W / 32] and their canonical statistics partials [M][W / 256][2] - what the FP8 residual GEMM's store
// pass (EPI_BIAS_RESID_LN8) leaves behind for every later layer, here for the embedding stage's rows (ln_pre's output).
// One wave per row; per 256-column segment a lane holds 4 consecutive columns: gemm.hpp ln8_row_segment. W % 256 == 0.
static __global__ void __launch_bounds__(256) rows_mx_stats_kernel(const float* __restrict__ x, unsigned char* __restrict__ x8,
                                                                   unsigned char* __restrict__ bs, float* __restrict__ part,
                                                                   int M, int W) {
    const int lane = threadIdx.x & 63;
    const long long r = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= M) return;
    const int nseg = W >> 8;
    for (int sg = 0; sg < nseg; ++sg) {
        const size_t col = (size_t)sg * 256 + lane * 4;
        const f32x4 v = *reinterpret_cast<const f32x4*>(x + (size_t)r * W + col);
        unsigned p4, sb;
        float sm, sq;
        ln8_row_segment(v, p4, sb, sm, sq);
        *reinterpret_cast<unsigned*>(x8 + (size_t)r * W + col) = p4;
        if ((lane & 7) == 0) bs[(size_t)r * (W >> 5) + (col >> 5)] = (unsigned char)sb;
        if (lane == 0) *reinterpret_cast<f32x2*>(part + ((size_t)r * nseg + sg) * 2) = f32x2{sm, sq};
    }
}

// ---------------------------------------------------------------------------------------------
// attention52_kernel: the ViT-B/32 shape (49 <= L <= 52, 4 key/query tiles, no mask) with the query tiles
// walked in a LOOP instead of all at once. attention_kernel<4,...> keeps 16 score tiles + 16 output tiles +
// 16 fragments live (~120 registers with the AGPRs: 4 waves per SIMD) and stages 64 V rows per wave (32 KiB per
// workgroup: 4 workgroups per CU); its lifetime per wave is ~20 us of mostly latency, so residency is its
// throughput (tools/attn_cliff.py: steps at every 16 waves per CU). Here per query tile: 4 score tiles, softmax,
// 4 output tiles, store; K fragments loaded once, the next tile's Q fragments in flight during the current tile,
// V staged once as 52 rows (6.5 KiB per wave -> 26 KiB per workgroup, six workgroups = 24 waves per CU); keys
// >= L carry zero weight and their transposed reads are redirected to row 51 (finite filler). Same arithmetic
// per (query, key, d) as attention_kernel: bit-identical output.
// ---------------------------------------------------------------------------------------------
static __global__ void __launch_bounds__(256, 6) attention52_kernel(const unsigned short* __restrict__ qkv,
                                                                    unsigned short* __restrict__ out, int B, int L,
                                                                    int heads) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NT = 4, ROWS = 52;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int fr = lane & 15, fg = lane >> 4;
    const int W = heads * 64;
    const long long item = (long long)blockIdx.x * 4 + wave;      // (b, h)
    if (item >= (long long)B * heads) return;                       // wave-uniform; no block barriers below
    const int b = (int)(item / heads), h = (int)(item - (long long)b * heads);
    const unsigned short* base = qkv + (size_t)b * L * 3 * W + h * 64;
    const size_t rs = (size_t)3 * W;
    char* vt = smem + wave * (ROWS * 128);

    // ---- stage V rows [0, 52): 8 rows x 128 B per wave-instruction, the last one half masked
#pragma unroll
    for (int i = 0; i < 7; ++i) {
        const int row = i * 8 + (lane >> 3);
        const int srcrow = row < L ? row : L - 1;                   // rows L..51: finite filler, weight 0
        const uint4 d = *reinterpret_cast<const uint4*>(base + (size_t)srcrow * rs + 2 * W + (lane & 7) * 8);
        if (row < ROWS) *reinterpret_cast<uint4*>(vt + row * 128 + (lane & 7) * 16) = d;
    }
    // ---- K fragments of all four key tiles (kept), Q fragments of tile 0
    bf16x8 kf[NT][2];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        int row = t * 16 + fr;
        row = row < L ? row : L - 1;
        const unsigned short* pr = base + (size_t)row * rs + W + fg * 8;
        kf[t][0] = *reinterpret_cast<const bf16x8*>(pr);
        kf[t][1] = *reinterpret_cast<const bf16x8*>(pr + 32);
    }
    auto q_ptr = [&](int qt) {
        int row = qt * 16 + fr;
        row = row < L ? row : L - 1;
        return base + (size_t)row * rs + fg * 8;
    };
    bf16x8 qn0 = *reinterpret_cast<const bf16x8*>(q_ptr(0));
    bf16x8 qn1 = *reinterpret_cast<const bf16x8*>(q_ptr(0) + 32);
    // transposed-read addresses of the two 32-key steps (see attention_kernel); key rows >= 52 -> row 51
    const int k00 = 4 * fg + (fr >> 2);
    const int krow[4] = {k00, k00 + 16, k00 + 32, (k00 + 48 < ROWS ? k00 + 48 : ROWS - 1)};
    const char* vcol = vt + (4 * (fr & 3)) * 2;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // this wave's V tile is in LDS
    __builtin_amdgcn_wave_barrier();

#pragma unroll 1
    for (int qt = 0; qt < NT; ++qt) {
        const bf16x8 q0 = qn0, q1 = qn1;
        if (qt + 1 < NT) {
            qn0 = *reinterpret_cast<const bf16x8*>(q_ptr(qt + 1));
            qn1 = *reinterpret_cast<const bf16x8*>(q_ptr(qt + 1) + 32);
        }
        f32x4 s[NT];
#pragma unroll
        for (int kt = 0; kt < NT; ++kt) {
            f32x4 a = {0.f, 0.f, 0.f, 0.f};
            a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[kt][0], q0, a, 0, 0, 0);
            a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[kt][1], q1, a, 0, 0, 0);
            s[kt] = a;
        }
        float mx = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < NT; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int ki = kt * 16 + 4 * fg + r;
                float x = s[kt][r] * 0.125f;
                if (ki >= L) x = -INFINITY;
                s[kt][r] = x;
                mx = fmaxf(mx, x);
            }
        mx = fmaxf(mx, __shfl_xor(mx, 16));
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        float sum = 0.f;
#pragma unroll
        for (int kt = 0; kt < NT; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float e = __expf(s[kt][r] - mx);             // key 0 is never masked: mx is finite
                s[kt][r] = e;
                sum += e;
            }
        sum += __shfl_xor(sum, 16);
        sum += __shfl_xor(sum, 32);
        const float inv = 1.0f / sum;
#pragma unroll
        for (int kt = 0; kt < NT; ++kt) s[kt] *= inv;

        f32x4 o[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) o[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const f32x4 lo = s[2 * ks], hi = s[2 * ks + 1];
            const uint4 u = make_uint4(pack_bf16x2(lo.x, lo.y), pack_bf16x2(lo.z, lo.w), pack_bf16x2(hi.x, hi.y),
                                       pack_bf16x2(hi.z, hi.w));
            const bf16x8 pf = __builtin_bit_cast(bf16x8, u);
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                const s16x4 t0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                    (__attribute__((address_space(3))) s16x4*)(vcol + krow[2 * ks] * 128 + dt * 32));
                const s16x4 t1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                    (__attribute__((address_space(3))) s16x4*)(vcol + krow[2 * ks + 1] * 128 + dt * 32));
                typedef short s16x8 __attribute__((ext_vector_type(8)));
                const s16x8 t = {t0.x, t0.y, t0.z, t0.w, t1.x, t1.y, t1.z, t1.w};
                o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, t), pf, o[dt], 0, 0, 0);
            }
        }
        const int qi = qt * 16 + fr;
        if (qi < L) {
            unsigned short* dst = out + ((size_t)b * L + qi) * W + h * 64 + 4 * fg;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt)
                *reinterpret_cast<uint2*>(dst + dt * 16) = make_uint2(pack_bf16x2(o[dt].x, o[dt].y), pack_bf16x2(o[dt].z, o[dt].w));
        }
    }
}

// ---------------------------------------------------------------------------------------------
// attention52x4_kernel: the ViT-B/32 attention (49 <= L <= 52, no mask) with ONE WORKGROUP per (image, head) and one
// 16-query tile per wave. attention52_kernel gives a whole (image, head) to one wave: 17 dependent-latency global loads
// in 64-byte pieces, then four query tiles one after the other - a wave lives ~20 us and the launch is bound by that
// residency (65 us per layer at B = 870 = 4.1 TB/s). Here the four waves stage K (64 rows, 16-byte chunks XOR-swizzled by
// row & 7 for conflict-free ds_read_b128 fragments) and V (52 rows, row-major for ds_read_b64_tr_b16) ONCE as whole
// 128-byte head rows, <= 4 loads per lane, one barrier, and each wave then runs exactly one iteration of
// attention52_kernel's query-tile loop: same MFMA sequence, same softmax: bit-identical output, a quarter of the serial
// chain. 14.5 KiB of LDS per workgroup, 8 workgroups (32 waves) per CU.
// ---------------------------------------------------------------------------------------------
template <bool OUT8>
static __global__ void __launch_bounds__(256, 5) attention52x4_kernel(const unsigned short* __restrict__ qkv,
                                                                      unsigned short* __restrict__ out, int B, int L,
                                                                      int heads, unsigned char* __restrict__ out8, unsigned char* __restrict__ out_bs) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NT = 4, ROWS = 52;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int fr = lane & 15, fg = lane >> 4;
    const int W = heads * 64;
    const size_t rs = (size_t)3 * W;
    char* kt_ = smem;                       // K: 64 rows x 128 B, swizzled
    char* vt = smem + 64 * 128;             // V: 52 rows x 128 B
    const int srow = lane >> 3, sch = lane & 7;
    const int items = B * heads;

    // A workgroup walks items blockIdx.x, + gridDim.x, ... (the grid is sized to the chip's residency): the NEXT item's six
    // loads per lane (Q fragments, K / V staging pieces) are issued before the current item's math, so that the HBM round
    // trip of item i + 1 runs under the MFMA / softmax / LDS work of item i instead of in front of it.
    // (a macro, not a lambda: by-reference captures of the register arrays put them in scratch)
    bf16x8 q0n, q1n;
    uint4 kdn0, kdn1, vdn0, vdn1;
#define ATT52_FETCH(IT)                                                                                  \
    do {                                                                                                 \
        const int b_ = (IT) / heads, h_ = (IT) - b_ * heads;                                             \
        const unsigned short* base_ = qkv + (size_t)b_ * L * 3 * W + h_ * 64;                            \
        int qrow_ = wave * 16 + fr;                                                                      \
        qrow_ = qrow_ < L ? qrow_ : L - 1;                                                               \
        const unsigned short* qp_ = base_ + (size_t)qrow_ * rs + fg * 8;                                 \
        q0n = *reinterpret_cast<const bf16x8*>(qp_);                                                     \
        q1n = *reinterpret_cast<const bf16x8*>(qp_ + 32);                                                \
        int r0_ = (wave * 2) * 8 + srow, r1_ = (wave * 2 + 1) * 8 + srow;   /* K rows 0..63, V rows 0..55 */ \
        r0_ = r0_ < L ? r0_ : L - 1;                                        /* rows >= L: finite filler */ \
        r1_ = r1_ < L ? r1_ : L - 1;                                                                     \
        kdn0 = *reinterpret_cast<const uint4*>(base_ + (size_t)r0_ * rs + W + sch * 8);                  \
        vdn0 = *reinterpret_cast<const uint4*>(base_ + (size_t)r0_ * rs + 2 * W + sch * 8);              \
        kdn1 = *reinterpret_cast<const uint4*>(base_ + (size_t)r1_ * rs + W + sch * 8);                  \
        vdn1 = *reinterpret_cast<const uint4*>(base_ + (size_t)r1_ * rs + 2 * W + sch * 8);              \
    } while (0)
    int item = blockIdx.x;
    if (item >= items) return;
    ATT52_FETCH(item);
    for (;;) {
        const int b = item / heads, h = item - b * heads;
        const bf16x8 q0 = q0n, q1 = q1n;
        {
            const int row0 = (wave * 2) * 8 + srow, row1 = row0 + 8;
            *reinterpret_cast<uint4*>(kt_ + row0 * 128 + ((sch ^ (row0 & 7)) << 4)) = kdn0;
            *reinterpret_cast<uint4*>(kt_ + row1 * 128 + ((sch ^ (row1 & 7)) << 4)) = kdn1;
            if (row0 < ROWS) *reinterpret_cast<uint4*>(vt + row0 * 128 + sch * 16) = vdn0;
            if (row1 < ROWS) *reinterpret_cast<uint4*>(vt + row1 * 128 + sch * 16) = vdn1;
        }
        const int nxt = item + gridDim.x;
        const bool has_next = nxt < items;
        if (has_next) ATT52_FETCH(nxt);
        __syncthreads();

    // K fragments of all four key tiles out of LDS: row t*16 + fr, logical chunks fg and 4 + fg
    bf16x8 kf[NT][2];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int row = t * 16 + fr;
        kf[t][0] = *reinterpret_cast<const bf16x8*>(kt_ + row * 128 + (((0 + fg) ^ (row & 7)) << 4));
        kf[t][1] = *reinterpret_cast<const bf16x8*>(kt_ + row * 128 + (((4 + fg) ^ (row & 7)) << 4));
    }
    const int k00 = 4 * fg + (fr >> 2);
    const int krow[4] = {k00, k00 + 16, k00 + 32, (k00 + 48 < ROWS ? k00 + 48 : ROWS - 1)};
    const char* vcol = vt + (4 * (fr & 3)) * 2;

    f32x4 s[NT];
#pragma unroll
    for (int kt = 0; kt < NT; ++kt) {
        f32x4 a = {0.f, 0.f, 0.f, 0.f};
        a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[kt][0], q0, a, 0, 0, 0);
        a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[kt][1], q1, a, 0, 0, 0);
        s[kt] = a;
    }
    float mx = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < NT; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int ki = kt * 16 + 4 * fg + r;
            float x = s[kt][r] * 0.125f;
            if (ki >= L) x = -INFINITY;
            s[kt][r] = x;
            mx = fmaxf(mx, x);
        }
    mx = fmaxf(mx, __shfl_xor(mx, 16));
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    float sum = 0.f;
#pragma unroll
    for (int kt = 0; kt < NT; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float e = __expf(s[kt][r] - mx);             // key 0 is never masked: mx is finite
            s[kt][r] = e;
            sum += e;
        }
    sum += __shfl_xor(sum, 16);
    sum += __shfl_xor(sum, 32);
    const float inv = 1.0f / sum;
#pragma unroll
    for (int kt = 0; kt < NT; ++kt) s[kt] *= inv;

    f32x4 o[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) o[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        const f32x4 lo = s[2 * ks], hi = s[2 * ks + 1];
        const uint4 u = make_uint4(pack_bf16x2(lo.x, lo.y), pack_bf16x2(lo.z, lo.w), pack_bf16x2(hi.x, hi.y),
                                   pack_bf16x2(hi.z, hi.w));
        const bf16x8 pf = __builtin_bit_cast(bf16x8, u);
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            const s16x4 t0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                (__attribute__((address_space(3))) s16x4*)(vcol + krow[2 * ks] * 128 + dt * 32));
            const s16x4 t1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                (__attribute__((address_space(3))) s16x4*)(vcol + krow[2 * ks + 1] * 128 + dt * 32));
            typedef short s16x8 __attribute__((ext_vector_type(8)));
            const s16x8 t = {t0.x, t0.y, t0.z, t0.w, t1.x, t1.y, t1.z, t1.w};
            o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, t), pf, o[dt], 0, 0, 0);
        }
    }
    // Output row qi, columns 16 dt + 4 fg .. + 3 per accumulator tile: as they stand that is four 8-byte stores per lane,
    // 32 contiguous bytes per row and instruction (store-issue-bound tail: the one-wave and the four-wave form took the
    // same 90 us from cold caches). Lanes fg and fg ^ 1 (lane ^ 16) swap halves - the even one gives away tiles 1, 3 and
    // receives the partner's 0, 2 - so that every lane owns 8 consecutive columns of two tiles: two 16-byte stores, 64
    // contiguous bytes per row and instruction.
    const int qi = wave * 16 + fr;
    uint2 pk[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) pk[dt] = make_uint2(pack_bf16x2(o[dt].x, o[dt].y), pack_bf16x2(o[dt].z, o[dt].w));
    const bool odd = fg & 1;
    uint2 mine[2], theirs[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const uint2 give = odd ? pk[2 * j] : pk[2 * j + 1];
        mine[j] = odd ? pk[2 * j + 1] : pk[2 * j];
        theirs[j] = make_uint2(__shfl_xor(give.x, 16), __shfl_xor(give.y, 16));
    }
    if (OUT8) {
        // FP8 towers: the rows leave as e4m3 with MX block scales (the A operand of out_proj, gemm256f8.hpp BSA) - the bytes
        // quantize_rows_fp8mx_kernel makes of the bf16 rows. A head's 64 columns are two 32-blocks; block j of row qi is
        // the 8 columns of tile 2 j (even fg) or 2 j + 1 (odd fg) in the four lanes fr, fr + 16, fr + 32, fr + 48.
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int dt = 2 * j + (odd ? 1 : 0);
            const uint4 v = odd ? make_uint4(theirs[j].x, theirs[j].y, mine[j].x, mine[j].y)
                                : make_uint4(mine[j].x, mine[j].y, theirs[j].x, theirs[j].y);
            const unsigned w_[4] = {v.x, v.y, v.z, v.w};
            float f[8];
            float mxa = 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                f[2 * i] = __uint_as_float(w_[i] << 16);
                f[2 * i + 1] = __uint_as_float(w_[i] & 0xffff0000u);
                mxa = fmaxf(mxa, fmaxf(fabsf(f[2 * i]), fabsf(f[2 * i + 1])));
            }
            mxa = fmaxf(mxa, __shfl_xor(mxa, 16));
            mxa = fmaxf(mxa, __shfl_xor(mxa, 32));
            const unsigned sb_ = fp8mx_scale_byte(mxa);
            const float inv_ = fp8mx_inv(sb_);
#pragma unroll
            for (int i = 0; i < 8; ++i) f[i] *= inv_;
            if (qi < L) {
                const size_t row_ = (size_t)b * L + qi;
                *reinterpret_cast<uint2*>(out8 + row_ * W + h * 64 + 4 * (fg & ~1) + dt * 16) = fp8_pack8(f);
                if (fg == 0) out_bs[row_ * (W >> 5) + h * 2 + j] = (unsigned char)sb_;
            }
        }
    } else if (qi < L) {
        // even fg: tiles 0, 2 at columns 16 dt + 4 fg .. + 7 (own 4, then the partner's); odd fg: tiles 1, 3 at 16 dt + 4 (fg - 1)
        unsigned short* dst = out + ((size_t)b * L + qi) * W + h * 64 + 4 * (fg & ~1);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int dt = 2 * j + (odd ? 1 : 0);
            const uint4 v = odd ? make_uint4(theirs[j].x, theirs[j].y, mine[j].x, mine[j].y)
                                : make_uint4(mine[j].x, mine[j].y, theirs[j].x, theirs[j].y);
            *reinterpret_cast<uint4*>(dst + dt * 16) = v;
        }
    }
        if (!has_next) break;
        item = nxt;
        __syncthreads();                    // every wave is done with this item's K / V tiles
    }
#undef ATT52_FETCH
}

// ---------------------------------------------------------------------------------------------
// Flash-style attention for long sequences (L > 80: ViT-B/16 L = 197, ViT-L/14 L = 257,
// ViT-L/14@336 L = 577), head dim 64, no mask. A workgroup of WPB waves owns one (sequence, head) and
// WPB consecutive 64-query blocks (one per wave); keys/values stream through in 64-key blocks that the
// workgroup stages ONCE into double-buffered LDS by LDS-DMA (issued before the block's math, one
// barrier per block), so K and V are read from L2 once per workgroup
// instead of once per query block. Online softmax in f32 with exp2 (scale and log2 e folded into one
// FMA). Same operand orientation as attention_kernel: S^T = K Q^T puts the key on the accumulator rows,
// so a block's P^T registers are the B operand of O^T += V^T P^T as they stand, and the per-query
// rescale 2^(m_old - m_new) multiplies whole accumulator tiles of this lane's column.
// K tile rows are 128 B with 16-byte chunks XOR-swizzled by row & 7 (conflict-free ds_read_b128 fragments);
// the V tile is row-major for ds_read_b64_tr_b16.
// ---------------------------------------------------------------------------------------------
// QT = 16-query tiles per wave: 2 (32 queries; 128 VGPRs, four waves per SIMD: with K/V staged once per workgroup either
// way, the smaller wave tile buys twice the resident waves of round 1's QT = 4 to overlap one wave's softmax with
// another's MFMAs). The block loop is VALU-bound (34 v_exp_f32 per 32 MFMAs per wave), so everything else was taken out
// of it: per-element masking selects (the last block is a separate instantiation), the staging addresses' integer
// multiplies (one per-lane base + uniform offsets), and the dependent MFMA pairs of S^T: 731 -> 437 instructions per block.
template <int WPB, int QT>
__global__ void __launch_bounds__(WPB * 64, QT == 2 ? 4 : 2) attention_flash_kernel(const unsigned short* __restrict__ qkv,
                                                                    unsigned short* __restrict__ out, int B, int L,
                                                                    int heads, int qgroups) {
    extern __shared__ __attribute__((aligned(16))) char smem[];      // 2 x [K 8 KiB | V 8 KiB]
    constexpr int NT = 4;          // key tiles of 16 per 64-key block
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fg = lane >> 4;
    const int W = heads * 64;
    const int qgi = blockIdx.x % qgroups;
    const int bh = blockIdx.x / qgroups;
    const int b = bh / heads, h = bh - b * heads;
    const unsigned short* base = qkv + (size_t)b * L * 3 * W + h * 64;
    const size_t rs = (size_t)3 * W;
    const int q0 = (qgi * WPB + wave) * (16 * QT);
    const bool wave_active = q0 < L;                  // idle waves still stage tiles and hit barriers

    bf16x8 qf[QT][2];
#pragma unroll
    for (int t = 0; t < QT; ++t) {
        int row = q0 + t * 16 + fr;
        row = row < L ? row : L - 1;
        const unsigned short* pr = base + (size_t)row * rs + fg * 8;
        qf[t][0] = *reinterpret_cast<const bf16x8*>(pr);
        qf[t][1] = *reinterpret_cast<const bf16x8*>(pr + 32);
    }
    f32x4 o[4][QT];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) o[dt][qt] = f32x4{0.f, 0.f, 0.f, 0.f};
    float m_run[QT], l_run[QT];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) { m_run[qt] = -INFINITY; l_run[qt] = 0.f; }

    // staging by LDS-DMA (no VGPRs): a K/V block is 16 pieces of 1 KiB (8 rows x 128 B); pieces 0..7 are
    // K (16-byte chunks XOR-swizzled with row & 7 — applied on the SOURCE address, the DMA destination is
    // lane-linear), pieces 8..15 are V (plain rows). Wave w issues pieces [w*PPW, (w+1)*PPW).
    // Source addresses: each lane keeps the address of its 16 bytes of every piece for key block 0 and adds the block's
    // offset (one 64-bit add per piece); only a block that reaches past row L - 1 recomputes clamped rows. (Recomputing
    // row * row_stride per piece and block cost 20 quarter-rate integer multiplies per block: with the masking selects
    // below, ~900 of the ~2900 VALU cycles of a block against 512 MFMA cycles.)
    constexpr int PPW = 16 / WPB;
    static_assert(PPW <= 8, "a wave's pieces are all K or all V");
    // this lane's 16 bytes of the wave's FIRST piece for key block 0; piece i is 8 rows further (row & 7 and with it the
    // swizzle stay the same), block k0 is k0 rows further
    const int pc0 = wave * PPW, isv0 = pc0 >> 3;
    const int row0 = (pc0 & 7) * 8 + (lane >> 3);
    const unsigned short* const src0 = base + (size_t)row0 * rs + (1 + isv0) * W + (isv0 ? (lane & 7) : ((lane & 7) ^ (row0 & 7))) * 8;
    auto issue_tiles = [&](int k0, int buf) {
        if (k0 + 64 <= L) {                                   // wave-uniform: every row of the block exists
#pragma unroll
            for (int i = 0; i < PPW; ++i) {
                const size_t uoff = (size_t)((unsigned)(k0 + 8 * i) * (unsigned)rs);      // uniform: scalar unit
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src0 + uoff),
                                                 (__attribute__((address_space(3))) void*)(smem + buf * 16384 + (pc0 + i) * 1024), 16, 0, 0);
            }
        } else {
            asm volatile("" ::: "memory");                    // keep the clamped form out of the common path
#pragma unroll
            for (int i = 0; i < PPW; ++i) {
                int row = k0 + row0 + 8 * i;
                row = row < L ? row : L - 1;                  // rows past the last one read the last one (masked below)
                const unsigned short* src = src0 + (size_t)((unsigned)(row - row0) * (unsigned)rs);
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                 (__attribute__((address_space(3))) void*)(smem + buf * 16384 + (pc0 + i) * 1024), 16, 0, 0);
            }
        }
    };
    issue_tiles(0, 0);

    const float c = 0.125f * 1.4426950408889634f;     // 1/sqrt(64) * log2(e)
    const int nb = (L + 63) >> 6;
    // one 64-key block; MASK (compile time) = the block reaches past key L - 1: only the last block can, so the loop over
    // the others carries no per-element selects (if-converted, they were 67 VALU instructions per block)
    auto block = [&](auto mask_tag, int j) {
        constexpr bool MASK = decltype(mask_tag)::value;
        const int k0 = j << 6;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's pieces of block j have landed
        __syncthreads();                                      // ... everyone's have; block j-1 is fully consumed
        if (j + 1 < nb) issue_tiles(k0 + 64, (j + 1) & 1);   // in flight under this block's math
        const char* kb = smem + (j & 1) * 16384;
        const char* vt = kb + 8192;
        if (wave_active) {
            // S^T tiles: the first halves (head dims 0..31) of all NT x QT accumulators, then the second halves, so that
            // no MFMA waits for the one before it (the two halves of one accumulator back to back did)
            f32x4 s[NT][QT];
#pragma unroll
            for (int hf = 0; hf < 2; ++hf)
#pragma unroll
                for (int kt = 0; kt < NT; ++kt) {
                    const int row = kt * 16 + fr;
                    const bf16x8 kf = *reinterpret_cast<const bf16x8*>(kb + row * 128 + (((4 * hf + fg) ^ (row & 7)) << 4));
#pragma unroll
                    for (int qt = 0; qt < QT; ++qt)
                        s[kt][qt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[qt][hf], hf ? s[kt][qt] : f32x4{0.f, 0.f, 0.f, 0.f},
                                                                            0, 0, 0);
                }
            if constexpr (MASK) {
#pragma unroll
                for (int qt = 0; qt < QT; ++qt)
#pragma unroll
                    for (int kt = 0; kt < NT; ++kt)
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            if (k0 + kt * 16 + 4 * fg + r >= L) s[kt][qt][r] = -INFINITY;
            }
#pragma unroll
            for (int qt = 0; qt < QT; ++qt) {
                float mx = -INFINITY;
#pragma unroll
                for (int kt = 0; kt < NT; ++kt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) mx = fmaxf(mx, s[kt][qt][r]);
                mx = fmaxf(mx, __shfl_xor(mx, 16));
                mx = fmaxf(mx, __shfl_xor(mx, 32));
                mx *= c;                                            // running max kept in the exp2 domain
                const float m_new = fmaxf(m_run[qt], mx);           // finite: every block holds >= 1 valid key
                const float alpha = __builtin_amdgcn_exp2f(m_run[qt] - m_new);   // 0 on the first block
                float sum = 0.f;
#pragma unroll
                for (int kt = 0; kt < NT; ++kt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float e = __builtin_amdgcn_exp2f(fmaf(s[kt][qt][r], c, -m_new));
                        s[kt][qt][r] = e;
                        sum += e;
                    }
                sum += __shfl_xor(sum, 16);
                sum += __shfl_xor(sum, 32);
                l_run[qt] = l_run[qt] * alpha + sum;
                m_run[qt] = m_new;
                // the running maximum of most rows stops moving after the first few key blocks: alpha is then exactly 1 and
                // the 16 multiplies per query tile are skipped wave-wide (x * 1 is exact: the same bits, 64 VALU cycles less
                // in a loop that is bound by its VALU issue, not by its 32 MFMAs)
                if (__ballot(alpha != 1.0f)) {
#pragma unroll
                    for (int dt = 0; dt < 4; ++dt) o[dt][qt] *= alpha;
                }
            }
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                bf16x8 pf[QT];
#pragma unroll
                for (int qt = 0; qt < QT; ++qt) {
                    const f32x4 lo = s[2 * ks][qt], hi = s[2 * ks + 1][qt];
                    const uint4 u = make_uint4(pack_bf16x2(lo.x, lo.y), pack_bf16x2(lo.z, lo.w), pack_bf16x2(hi.x, hi.y),
                                               pack_bf16x2(hi.z, hi.w));
                    pf[qt] = __builtin_bit_cast(bf16x8, u);
                }
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) {
                    const int kk = 32 * ks + 4 * fg + (fr >> 2);
                    const char* ad = vt + kk * 128 + (dt * 16 + 4 * (fr & 3)) * 2;
                    const s16x4 t0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(ad));
                    const s16x4 t1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(ad + 16 * 128));
                    typedef short s16x8 __attribute__((ext_vector_type(8)));
                    const s16x8 t = {t0.x, t0.y, t0.z, t0.w, t1.x, t1.y, t1.z, t1.w};
                    const bf16x8 vf = __builtin_bit_cast(bf16x8, t);
#pragma unroll
                    for (int qt = 0; qt < QT; ++qt)
                        o[dt][qt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf[qt], o[dt][qt], 0, 0, 0);
                }
            }
        }
    };
    for (int j = 0; j + 1 < nb; ++j) block(std::false_type{}, j);
    if ((nb << 6) > L) block(std::true_type{}, nb - 1);
    else block(std::false_type{}, nb - 1);

    if (!wave_active) return;
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
        const int qi = q0 + qt * 16 + fr;
        if (qi >= L) continue;
        const float inv = 1.0f / l_run[qt];
        unsigned short* dst = out + ((size_t)b * L + qi) * W + h * 64 + 4 * fg;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            const f32x4 v = o[dt][qt] * inv;
            *reinterpret_cast<uint2*>(dst + dt * 16) = make_uint2(pack_bf16x2(v.x, v.y), pack_bf16x2(v.z, v.w));
        }
    }
}

// host launchers (vit_kernels.hip)
int launch_layernorm(const LnArgs& a, hipStream_t st);
// out8 / out_bs / fused (FP8 towers): when the shape's kernel has the fused form, the rows are written as e4m3 + MX block
// scales into out8 / out_bs INSTEAD of bf16 into out and *fused is set; otherwise bf16 into out as always
int launch_attention(const unsigned short* qkv, unsigned short* out, int B, int L, int heads, int causal, int tr,
                     hipStream_t st, unsigned char* out8 = nullptr, unsigned char* out_bs = nullptr, bool* fused = nullptr,
                     hipEvent_t* probe_ev = nullptr);     // probe_ev[0..1]: the dispatch's own begin / end (bench probe)
int launch_patchify(const PatchArgs& a, hipStream_t st);
int launch_quantize_rows_fp8(const unsigned short* in, unsigned char* out, float* scale, int M, int K, hipStream_t st);
int launch_quantize_rows_fp8mx(const unsigned short* in, unsigned char* out, unsigned char* bscale, int M, int K, hipStream_t st);
int launch_rows_mx_stats(const float* x, unsigned char* x8, unsigned char* bs, float* part, int M, int W, hipStream_t st);

}  // namespace clipmi
