// jpeg.hip — `Image.open(tfn)` on the device for baseline JPEG files (the decode in front of
// `transform(...)`, reference build-index.py:47-48). Pillow decodes with libjpeg-turbo (third-party, absent from the
// reference tree); its published algorithm is restated here for the subset the host parser (cli-p_amd/jpeg.py) lets
// through - 8-bit baseline / extended-sequential Huffman, one interleaved scan, grey or YCbCr with luma sampling 1x1 /
// 2x1 / 2x2 and 1x1 chroma, with or without restart intervals - and produces Pillow's bytes (tests/test_jpeg_gpu.py compares with
// Pillow itself; oracle/jpeg_oracle.py is the CPU restatement, pinned against Pillow in tests/test_jpeg.py).
// Everything else (progressive, CMYK, PNG, ...) stays with Pillow in the decode workers.
//
// Kernels, one batch of images per call:
//   jpeg_build_luts_kernel   canonical Huffman codes of the batch's distinct DHT tables -> 10-bit look-up + slow-path arrays
//   jpeg_huffman_kernel      the entropy-coded segment -> int16 coefficients (DC still differential). Huffman decoding is a
//                            serial chain (a symbol's first bit is known only when the previous symbol has been decoded);
//                            a workgroup owns an image and decodes it in 1024-bit SUBSEQUENCES, one per thread: every thread
//                            starts at its subsequence's first bit as if a block began there, then repeatedly restarts from
//                            its predecessor's exit state (bit position, block within the MCU, coefficient index) until no
//                            exit state changes. Thread 0's start is exact, so by induction the fixed point is the
//                            sequential decode; Huffman streams re-synchronise by themselves after a few symbols, so on
//                            photographs the loop ends after 2-3 rounds (Klein & Wiseman's observation, used for JPEG by
//                            Weissenberger & Schmidt); on noise images without end-of-block symbols it degenerates to the
//                            serial chain, one subsequence per round, and still ends. A scan of the blocks completed per
//                            subsequence gives every thread its first block; a last pass writes the coefficients.
//                            Files with restart intervals need none of that: every interval starts on a byte the host found,
//                            at a known block, with fresh DC predictions - one thread walks one interval.
//   jpeg_dc_kernel           DC prediction: per-component running sums over the blocks in scan order
//   jpeg_idct_kernel         jidctint.c's jpeg_idct_islow with the dequantisation folded in, one block per thread
//   jpeg_color_kernel        jdsample.c's fancy (triangle) upsampling with jdmainct.c's edge rows + jdcolor.c's
//                            16-bit fixed-point YCbCr -> RGB; rows of width*3 bytes, as Pillow's array
#include "common.hpp"

namespace clipmi {
namespace {

struct JpegImage {             // mirrors clipmi_jpeg_image (include/clipmi.h)
    long long stream_off, coef_off, out_off, intervals_off;
    int stream_bytes;
    int width, height;
    int ncomp;
    int hs, vs;
    int dc_tbl[3], ac_tbl[3];
    int restart_interval, n_intervals;
    int stuffed, reserved;
    unsigned char quant[3][64];
};

constexpr int JP_T = 128;          // threads per workgroup = subsequences per chunk: ONE wave (waves that walk serial chains
                                   // must not share a SIMD: 870 two-wave workgroups ran 3.2 x slower than one)
constexpr int JP_SUB_BITS = 1024;  // bits per subsequence (32 words)
constexpr int JP_FAST = 10;        // bits of the direct look-up
constexpr int JP_SPEC = 3;         // speculative rounds before the rounds turn serial (jpeg_huffman_kernel)
constexpr int JP_RAW = 288;        // bytes of a raw table: 16 counts + 256 symbols + class (0 DC, 1 AC) + 15 zero bytes

// A 16-bit table entry says everything the state machine needs about a symbol: the bits to skip (code + the value bits behind it,
// 1..31), the step of the coefficient index (DC: 0 -> 1; AC: run + 1, 16 for ZRL, 64 = end of block) and the number of value bits.
// (LDS per workgroup decides how many images are resident at once, and an image can be a serial chain of milliseconds.)
__device__ __forceinline__ unsigned jp_entry(unsigned sym, unsigned len, bool ac) {
    const unsigned s = sym & 15, r = sym >> 4;
    const unsigned dz = !ac ? 1u : (s ? r + 1 : (r == 15 ? 16u : 64u));
    return (len + s) | (dz << 5) | (s << 12);
}
#define JP_ADV(e) ((e) & 31u)
#define JP_DZ(e) (((e) >> 5) & 127u)
#define JP_S(e) ((e) >> 12)

constexpr int JP_LONG = 8;         // canonical codes grow upwards, so the codes longer than JP_FAST bits sit under the LAST few
                                   // JP_FAST-bit prefixes (the standard tables: 5); the last JP_LONG prefixes get a 16-bit table
struct JpLut {
    unsigned short fast[1 << JP_FAST];   // jp_entry of codes of up to JP_FAST bits, 0 = longer (or invalid)
    unsigned short second[JP_LONG << (16 - JP_FAST)];   // by bits JP_FAST..15 for the last JP_LONG prefixes, 0 = invalid
    int maxcode[18];                     // largest code of length l (-1: none)
    int valoff[18];                      // symbol index of a code of length l = code + valoff[l]
    unsigned char vals[256];
    int covered;                         // every longer code sits in `second` (else: the bit-by-bit walk)
    int pad_[3];
};
static_assert(sizeof(JpLut) % 16 == 0, "tables are copied as 16-byte pieces");

__device__ const unsigned char jp_natural[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                                                 41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                                                 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

// Byte stuffing (a 0x00 behind every 0xFF of entropy-coded data) removed on the device, in place: the host then only slices the
// file (removing it there is 1.5 ms per MB of Python in a decode worker - the bound for photo-sized files). A workgroup per image
// walks the segment in 8-KB tiles: 32 bytes per thread in registers, kept bytes counted, an exclusive scan over the workgroup, the
// bytes written back at their compacted positions (never in front of a byte still to be read: the tile is in registers, and the
// output only falls behind the input). Any 0xFF followed by something else than 0x00 is a marker inside the scan: the record is
// tagged (stuffed = 2) and jpeg_huffman_kernel reports the file (status 3: Pillow decides).
__global__ void __launch_bounds__(256) jpeg_unstuff_kernel(unsigned char* __restrict__ streams, JpegImage* __restrict__ images) {
    JpegImage& im = images[blockIdx.x];
    if (im.stuffed != 1) return;
    __shared__ int wsum[4];
    __shared__ unsigned char lastb[256];
    __shared__ int s_marker;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    unsigned char* s = streams + im.stream_off;
    const int nraw = im.stream_bytes;
    if (tid == 0) s_marker = 0;
    int outpos = 0;
    unsigned char tile_prev = 0;
    for (int t0 = 0; t0 < nraw; t0 += 8192) {
        const int b0 = t0 + tid * 32;
        uint4 v[2] = {make_uint4(0, 0, 0, 0), make_uint4(0, 0, 0, 0)};
        if (b0 < nraw) v[0] = *reinterpret_cast<const uint4*>(s + b0);                 // (the segment is 16-byte aligned and padded)
        if (b0 + 16 < nraw) v[1] = *reinterpret_cast<const uint4*>(s + b0 + 16);
        const unsigned char* b = reinterpret_cast<const unsigned char*>(v);
        lastb[tid] = b[31];
        __syncthreads();
        unsigned char prev = tid ? lastb[tid - 1] : tile_prev;
        const unsigned char next_prev = lastb[255];
        int nk = 0, mk = 0;
        unsigned dropmask = 0;                              // bit k: byte k is a stuffed 0x00 (or lies behind the segment)
#pragma unroll
        for (int k = 0; k < 32; ++k) {
            const unsigned char cur = b[k];
            const bool in = b0 + k < nraw;
            const bool drop = !in || (prev == 0xFF && cur == 0);
            if (in && prev == 0xFF && cur != 0) mk = 1;
            if (in && b0 + k == nraw - 1 && cur == 0xFF) mk = 1;         // a 0xFF with nothing behind it
            dropmask |= (drop ? 1u : 0u) << k;
            nk += drop ? 0 : 1;
            prev = cur;
        }
        // exclusive scan of nk over the workgroup: within the wave by shuffles, across the four waves through LDS
        int inc = nk;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int o = __shfl_up(inc, d);
            if (lane >= d) inc += o;
        }
        if (lane == 63) wsum[wave] = inc;
        __syncthreads();
        int base = outpos;
        for (int w = 0; w < wave; ++w) base += wsum[w];
        const int total = wsum[0] + wsum[1] + wsum[2] + wsum[3];
        unsigned char* o = s + base + inc - nk;
#pragma unroll
        for (int k = 0; k < 32; ++k)
            if (!((dropmask >> k) & 1)) *o++ = b[k];
        if (mk) s_marker = 1;
        outpos += total;
        tile_prev = next_prev;
        __syncthreads();                                    // wsum / lastb are rewritten by the next tile
    }
    if (tid < 16) s[outpos + tid] = 0;                      // a zero tail for the last word (>= 16 bytes of padding follow the segment)
    __syncthreads();
    if (tid == 0) {
        im.stream_bytes = outpos;
        im.stuffed = s_marker ? 2 : 0;
    }
}

__global__ void __launch_bounds__(256) jpeg_build_luts_kernel(const unsigned char* __restrict__ raw, JpLut* __restrict__ luts) {
    __shared__ int first_code[18], first_idx[18];
    const unsigned char* t = raw + (size_t)blockIdx.x * JP_RAW;
    JpLut& L = luts[blockIdx.x];
    const int tid = threadIdx.x;
    for (int i = tid; i < (1 << JP_FAST); i += 256) L.fast[i] = 0;
    for (int i = tid; i < (JP_LONG << (16 - JP_FAST)); i += 256) L.second[i] = 0;
    if (tid == 0) {
        int code = 0, k = 0, covered = 1;
        for (int l = 1; l <= 16; ++l) {
            const int n = t[l - 1];
            first_code[l] = code;
            first_idx[l] = k;
            L.maxcode[l] = n ? code + n - 1 : -1;
            L.valoff[l] = k - code;
            if (n && l > JP_FAST && (code >> (l - JP_FAST)) < (1 << JP_FAST) - JP_LONG) covered = 0;
            code = (code + n) << 1;
            k += n;
        }
        first_idx[17] = k > 256 ? 256 : k;
        L.maxcode[0] = L.maxcode[17] = -1;
        L.valoff[0] = L.valoff[17] = 0;
        L.covered = covered;
    }
    L.vals[tid] = t[16 + tid];
    __syncthreads();
    if (tid < first_idx[17]) {
        int l = 1;
        while (l < 16 && tid >= first_idx[l + 1]) ++l;
        const int code = first_code[l] + (tid - first_idx[l]);
        const unsigned short e = (unsigned short)jp_entry(t[16 + tid], (unsigned)l, t[272] != 0);
        if (l <= JP_FAST && code < (1 << l)) {
            const int base = code << (JP_FAST - l);
            for (int j = 0; j < (1 << (JP_FAST - l)); ++j) L.fast[base + j] = e;
        } else if (l > JP_FAST && code < (1 << l)) {
            const int v16 = code << (16 - l);
            const int pre = (v16 >> (16 - JP_FAST)) - ((1 << JP_FAST) - JP_LONG);
            if (pre >= 0) {
                const int base = (pre << (16 - JP_FAST)) + (v16 & ((1 << (16 - JP_FAST)) - 1));
                for (int j = 0; j < (1 << (16 - l)); ++j) L.second[base + j] = e;
            }
        }
    }
}

struct JpState {
    unsigned p;      // bit position in the image's entropy-coded segment
    unsigned bz;     // (block within the MCU) << 8 | index of the next coefficient (0 = the block's DC symbol comes next)
};

struct JpShared {
    JpLut lut[6];                              // [component][DC, AC]
    unsigned words[JP_T * 33 + 40];            // the chunk's big-endian words, word i at i + i / 32 (subsequence i at 33 i: bank spread)
    JpState exit_state[JP_T];
    JpState start_state[JP_T];
    int blocks[JP_T];
    unsigned char nat[64];
    int first[2];                              // (restart intervals: [0] = an interval ended early)
    int end_p;
    int bad;
};

// The table entry of the symbol whose code starts the 32-bit window x (0: no such code)
__device__ __forceinline__ unsigned jp_lookup(const JpLut& L, unsigned x, bool ac) {
    const unsigned k = x >> (32 - JP_FAST);
    unsigned e = L.fast[k];
    if (!e) {
        if (k >= (1u << JP_FAST) - JP_LONG) {
            e = L.second[((k - ((1u << JP_FAST) - JP_LONG)) << (16 - JP_FAST)) + ((x >> 16) & ((1u << (16 - JP_FAST)) - 1))];
        } else if (!L.covered) {
            for (int l = JP_FAST + 1; l <= 16; ++l) {
                const int code = (int)(x >> (32 - l));
                if (code <= L.maxcode[l]) {
                    e = jp_entry(L.vals[(code + L.valoff[l]) & 255], (unsigned)l, ac);
                    break;
                }
            }
        }
    }
    return e;
}

__device__ __forceinline__ int jp_comp(unsigned blk, int hv) { return blk < (unsigned)hv ? 0 : (int)blk - hv + 1; }

// One subsequence: symbols from state (p, bz) while p < end. WRITE: also stores the coefficients of blocks ablk, ablk+1, ...
// (stops behind block `total` - 1). Returns the number of blocks completed.
// GLOBAL: the words come straight from the image's segment in memory (restart intervals: a thread walks a whole interval).
template <bool WRITE, bool GLOBAL = false>
__device__ __forceinline__ int jp_decode(JpShared& sh, unsigned cw0, int bpm, int hv, unsigned& p, unsigned& bz, unsigned end,
                                         short* __restrict__ coef, long long ablk, long long total,
                                         const unsigned* __restrict__ src = nullptr, unsigned nwords = 0) {
    unsigned blk = bz >> 8, z = bz & 255;
    int done = 0;
    unsigned cur = ~0u, w0 = 0, w1 = 0;
    int comp = blk < (unsigned)hv ? 0 : (int)blk - hv + 1;
    while (p < end) {
        const unsigned gw = p >> 5;
        if (gw != cur) {
            cur = gw;
            if (GLOBAL) {
                w0 = gw < nwords ? __builtin_bswap32(src[gw]) : 0u;
                w1 = gw + 1 < nwords ? __builtin_bswap32(src[gw + 1]) : 0u;
            } else {
                const unsigned r = gw - cw0;
                w0 = sh.words[r + (r >> 5)];               // (word i sits at i + i / 32)
                w1 = sh.words[r + 1 + ((r + 1) >> 5)];
            }
        }
        const unsigned x = (unsigned)(((((unsigned long long)w0) << 32) | w1) << (p & 31) >> 32);
        const bool ac = z != 0;
        const JpLut& L = sh.lut[comp * 2 + (ac ? 1 : 0)];
        const unsigned e = jp_lookup(L, x, ac);
        if (!e) {                            // no such code: a wrong start state's garbage (discarded) or a corrupt file
            if (WRITE) sh.bad = 1;
            p += 1;
            continue;
        }
        if (WRITE) {
            const unsigned s = JP_S(e);
            int v = 0;
            if (s) {
                v = (int)((x << (JP_ADV(e) - s)) >> (32 - s));
                if (v < (1 << (s - 1))) v -= (1 << s) - 1;
            }
            if (!ac) {
                coef[ablk * 64] = (short)v;
            } else if (s) {
                const unsigned k = z + JP_DZ(e) - 1;
                coef[ablk * 64 + (k < 64 ? sh.nat[k] : 63)] = (short)v;
            }
        }
        p += JP_ADV(e);
        z += JP_DZ(e);
        if (z >= 64) {
            z = 0;
            blk = blk + 1 == (unsigned)bpm ? 0 : blk + 1;
            comp = blk < (unsigned)hv ? 0 : (int)blk - hv + 1;
            ++done;
            if (WRITE) {
                ++ablk;
                if (ablk >= total) {
                    sh.end_p = (int)p;
                    break;
                }
            }
        }
    }
    bz = (blk << 8) | z;
    return done;
}

// The same walk as jp_decode<false>, by a whole WAVE for ONE subsequence (all arguments wave-uniform): when only one or two
// threads of a wave have anything to re-decode - always, on images whose subsequences do not re-synchronise - a lane's serial
// chain of look-ups (window -> LDS -> length -> next window, ~470 cycles per symbol) is the whole kernel's time. Here the 64
// lanes look up the entries of the 64 bit offsets behind p at once, and the chain that is left is a scalar walk over registers:
// v_readlane of the entry at the current offset, add its bit count, step the coefficient index. A window ends after 64 bits or
// with its block (the next block may use other tables).
constexpr int JP_WIN = 4;          // 64-bit pieces of a wave's window: the look-ups of all pieces are in flight together, so the
                                   // two LDS round trips in front of a walk are paid once per JP_WIN * 64 bits
__device__ __forceinline__ int jp_decode_wave(JpShared& sh, unsigned cw0, int bpm, int hv, unsigned& p_, unsigned& bz_, unsigned end) {
    const unsigned lane = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    unsigned p = p_, blk = bz_ >> 8, z = bz_ & 255;
    int done = 0;
    while (p < end) {
        const int c = jp_comp(blk, hv);
        unsigned x[JP_WIN], a[JP_WIN];
#pragma unroll
        for (int h = 0; h < JP_WIN; ++h) {
            const unsigned q = p + 64u * h + lane;
            unsigned r = (q >> 5) - cw0;
            r = r < JP_T * 32u + 6u ? r : JP_T * 32u + 6u;                            // (behind the staged words: never walked)
            const unsigned w0 = sh.words[r + (r >> 5)], w1 = sh.words[r + 1 + ((r + 1) >> 5)];    // (word i sits at i + i / 32)
            x[h] = (unsigned)(((((unsigned long long)w0) << 32) | w1) << (q & 31) >> 32);
        }
#pragma unroll
        for (int h = 0; h < JP_WIN; ++h) {
            const unsigned e = jp_lookup(sh.lut[2 * c + 1], x[h], true);
            a[h] = e ? JP_ADV(e) | (JP_DZ(e) << 16) : 1u;                // (no such code: one bit on, no step - as jp_decode)
        }
        const unsigned room = end - p < 64u * JP_WIN ? end - p : 64u * JP_WIN;
        unsigned o = 0;
        if (z == 0) {                                    // the block's DC symbol starts the window
            const unsigned e = __builtin_amdgcn_readfirstlane(jp_lookup(sh.lut[2 * c], x[0], false));
            if (!e) {
                p += 1;
                continue;
            }
            o = JP_ADV(e);
            z = 1;
        }
        // AC symbols up to the window's or the block's end, piece by piece. The walk's state is ONE register - the offset, biased
        // so that bit 15 comes up at the piece's limit, and the coefficient index from bit 16, where 64 is bit 22 - and an entry is
        // (bits to skip | index step << 16): a symbol is a v_readlane, an add and a test.
#pragma unroll
        for (int h = 0; h < JP_WIN; ++h) {
            const unsigned lim = room < 64u * (h + 1) ? room : 64u * (h + 1);
            if ((o < lim) & (z < 64)) {
                const unsigned bias = 0x8000u - lim;
                unsigned st = (o + bias) | (z << 16);
                do {
                    st += __builtin_amdgcn_readlane(a[h], (st & 0x7fffu) - bias - 64u * h);
                } while (!(st & 0x00c08000u));
                o = (st & 0xffffu) - bias;
                z = st >> 16;
            }
        }
        if (z >= 64) {
            z = 0;
            blk = blk + 1 == (unsigned)bpm ? 0 : blk + 1;
            ++done;
        }
        p += o;
    }
    p_ = p;
    bz_ = (blk << 8) | z;
    return done;
}

__global__ void __launch_bounds__(JP_T) jpeg_huffman_kernel(const unsigned char* __restrict__ streams, const JpegImage* __restrict__ images,
                                                           const JpLut* __restrict__ luts, short* __restrict__ coef_all,
                                                           int* __restrict__ status) {
    __shared__ JpShared sh;
    const JpegImage& im = images[blockIdx.x];
    const int tid = threadIdx.x;
    const int hv = im.ncomp == 1 ? 1 : im.hs * im.vs;
    const int bpm = im.ncomp == 1 ? 1 : hv + 2;
    const long long mx = (im.width + 8 * (im.ncomp == 1 ? 1 : im.hs) - 1) / (8 * (im.ncomp == 1 ? 1 : im.hs));
    const long long my = (im.height + 8 * (im.ncomp == 1 ? 1 : im.vs) - 1) / (8 * (im.ncomp == 1 ? 1 : im.vs));
    const long long total = mx * my * bpm;
    const unsigned nbits = (unsigned)im.stream_bytes * 8u;
    short* coef = coef_all + im.coef_off * 64;
    const unsigned* src = reinterpret_cast<const unsigned*>(streams + im.stream_off);
    const unsigned nwords = ((unsigned)im.stream_bytes + 3u) >> 2;

    for (int c = 0; c < im.ncomp; ++c) {
        const uint4* d = reinterpret_cast<const uint4*>(&luts[im.dc_tbl[c]]);
        const uint4* a = reinterpret_cast<const uint4*>(&luts[im.ac_tbl[c]]);
        uint4* dd = reinterpret_cast<uint4*>(&sh.lut[2 * c]);
        uint4* da = reinterpret_cast<uint4*>(&sh.lut[2 * c + 1]);
        for (int i = tid; i < (int)(sizeof(JpLut) / 16); i += JP_T) {
            dd[i] = d[i];
            da[i] = a[i];
        }
    }
    if (tid < 64) sh.nat[tid] = jp_natural[tid];
    if (tid == 0) {
        sh.end_p = -1;
        sh.bad = 0;
    }
    if (im.stuffed == 2) {                              // jpeg_unstuff_kernel found a marker inside the scan
        if (tid == 0) status[blockIdx.x] = 3;
        return;
    }
    if (im.restart_interval > 0) {
        // Restart intervals: every interval starts on a byte the host knows, with fresh DC predictions, and its first block is
        // k * interval * blocks per MCU - independent chains, one thread each, no synchronisation to find.
        if (tid == 0) sh.first[0] = 0;
        __syncthreads();
        const unsigned* offs = reinterpret_cast<const unsigned*>(streams + im.intervals_off);
        const long long nmcu = mx * my, ri = im.restart_interval;
        for (int k = tid; k < im.n_intervals; k += JP_T) {
            unsigned p = offs[k] * 8u, bz = 0;
            const unsigned end = (k + 1 < im.n_intervals ? offs[k + 1] : (unsigned)im.stream_bytes) * 8u;
            const long long left = nmcu - k * ri, want = (left < ri ? left : ri) * bpm;
            const int done = end <= nbits && p <= end
                                 ? jp_decode<true, true>(sh, 0, bpm, hv, p, bz, end, coef, k * ri * bpm, k * ri * bpm + want, src, nwords)
                                 : -1;
            if (done != want || p > end) sh.first[0] = 1;          // the interval's data ended early or ran over
        }
        __syncthreads();
        if (tid == 0) status[blockIdx.x] = sh.bad ? 1 : (sh.first[0] || (long long)im.n_intervals * ri < nmcu ? 2 : 0);
        return;
    }
    JpState carry{0u, 0u};
    long long carry_blocks = 0;
    const unsigned nsub = (nbits + JP_SUB_BITS - 1) / JP_SUB_BITS;
    for (unsigned c0 = 0; c0 < nsub && carry_blocks < total; c0 += JP_T) {
        __syncthreads();                                   // the previous chunk's readers are done with sh.words
        const unsigned cw0 = c0 * 32u;
        for (unsigned r = tid; r < (unsigned)JP_T * 32u + 8u; r += JP_T) {      // the chunk's words + 8 of the next chunk
            const unsigned g = cw0 + r;
            sh.words[r + (r >> 5)] = g < nwords ? __builtin_bswap32(src[g]) : 0u;
        }
        const unsigned i = c0 + tid;
        const bool act = i < nsub;
        const unsigned end = (i + 1) * JP_SUB_BITS < nbits ? (i + 1) * JP_SUB_BITS : nbits;
        JpState start = tid ? JpState{i * JP_SUB_BITS, 0u} : carry;
        JpState ex = start;
        int nb = 0;
        __syncthreads();
        if (act) nb = jp_decode<false>(sh, cw0, bpm, hv, ex.p, ex.bz, end, nullptr, 0, 0);
        sh.exit_state[tid] = ex;
        // Rounds. JP_SPEC speculative ones: every thread whose predecessor's exit state moved decodes again - on photographs that
        // is all it takes. When states still move after that (noise: nothing re-synchronises), wave 0 walks the rest of the chunk
        // ALONE, one subsequence after the other from the first thread whose predecessor moved (jp_decode_wave; a subsequence whose
        // start did not move is skipped): no barrier, no exchange per subsequence - everything in front of that thread is final,
        // everything behind it would decode garbage, and a round per subsequence cost more than its walk. Either way the fixed
        // point is the sequential decode: thread 0 is exact, and a thread that decodes from a final predecessor is final.
        bool moving = true;
        for (int round = 0; round < JP_SPEC && moving; ++round) {
            __syncthreads();
            const JpState prev = tid ? sh.exit_state[tid - 1] : carry;
            const bool ch = act && tid && (prev.p != start.p || prev.bz != start.bz);
            moving = __syncthreads_or(ch);
            if (!moving) break;
            const unsigned long long m = __ballot(ch);
            if (__popcll(m) <= 2) {                        // (wave-uniform) the whole wave walks each of them: jp_decode_wave
                unsigned long long mm = m;
                while (mm) {
                    const int b = __builtin_ctzll(mm);
                    mm &= mm - 1;
                    unsigned sp = __builtin_amdgcn_readlane(prev.p, b), sbz = __builtin_amdgcn_readlane(prev.bz, b);
                    const unsigned se = __builtin_amdgcn_readlane(end, b);
                    const int d = jp_decode_wave(sh, cw0, bpm, hv, sp, sbz, se);
                    if ((tid & 63) == b) {
                        start = prev;
                        ex = JpState{sp, sbz};
                        nb = d;
                        sh.exit_state[tid] = ex;
                    }
                }
            } else if (ch) {
                start = prev;
                ex = start;
                nb = jp_decode<false>(sh, cw0, bpm, hv, ex.p, ex.bz, end, nullptr, 0, 0);
                sh.exit_state[tid] = ex;
            }
        }
        if (moving) {
            sh.start_state[tid] = start;
            sh.blocks[tid] = nb;
            __syncthreads();
            if (tid < 64) {
                const unsigned nact = nsub - c0 < (unsigned)JP_T ? nsub - c0 : (unsigned)JP_T;
                JpState run = carry;
                bool fresh = false;                       // `run` comes from a subsequence this walk decoded
                for (unsigned i = 1; i < nact; ++i) {
                    const JpState pe = fresh ? run : sh.exit_state[i - 1], st = sh.start_state[i];
                    if (pe.p == st.p && pe.bz == st.bz) {  // consistent with its predecessor: its exit state stands
                        fresh = false;
                        continue;
                    }
                    unsigned sp = __builtin_amdgcn_readfirstlane(pe.p), sbz = __builtin_amdgcn_readfirstlane(pe.bz);    // (wave-uniform)
                    const unsigned se = (c0 + i + 1) * JP_SUB_BITS < nbits ? (c0 + i + 1) * JP_SUB_BITS : nbits;
                    const int d = jp_decode_wave(sh, cw0, bpm, hv, sp, sbz, se);
                    if (tid == 0) {
                        sh.start_state[i] = pe;
                        sh.exit_state[i] = JpState{sp, sbz};
                        sh.blocks[i] = d;
                    }
                    run = JpState{sp, sbz};
                    fresh = true;
                }
            }
            __syncthreads();
            start = sh.start_state[tid];
            ex = sh.exit_state[tid];
            nb = sh.blocks[tid];
            __syncthreads();                               // sh.blocks is rewritten below
        }
        // first block of every subsequence: exclusive scan of the blocks completed
        sh.blocks[tid] = act ? nb : 0;
        __syncthreads();
        for (int d = 1; d < JP_T; d <<= 1) {
            const int add = tid >= d ? sh.blocks[tid - d] : 0;
            __syncthreads();
            sh.blocks[tid] += add;
            __syncthreads();
        }
        const long long first = carry_blocks + sh.blocks[tid] - (act ? nb : 0);
        if (act && first < total) {
            JpState w = start;
            jp_decode<true>(sh, cw0, bpm, hv, w.p, w.bz, end, coef, first, total);
        }
        carry_blocks += sh.blocks[JP_T - 1];
        const unsigned last = (nsub - c0 < (unsigned)JP_T ? nsub - c0 : (unsigned)JP_T) - 1;
        carry = sh.exit_state[last];
    }
    __syncthreads();
    if (tid == 0) {
        int st = 0;
        if (sh.bad) st = 1;
        else if (carry_blocks < total || sh.end_p < 0 || (unsigned)sh.end_p > nbits) st = 2;
        status[blockIdx.x] = st;
    }
}

// DC prediction (jdhuff.c: last_dc_val per component): running sums over the blocks in scan order
__global__ void __launch_bounds__(256) jpeg_dc_kernel(const JpegImage* __restrict__ images, short* __restrict__ coef_all) {
    __shared__ int part[3][256];
    const JpegImage& im = images[blockIdx.x];
    const int tid = threadIdx.x;
    const int hs = im.ncomp == 1 ? 1 : im.hs, vs = im.ncomp == 1 ? 1 : im.vs;
    const int hv = hs * vs, bpm = im.ncomp == 1 ? 1 : hv + 2;
    const long long nmcu = (long long)((im.width + 8 * hs - 1) / (8 * hs)) * ((im.height + 8 * vs - 1) / (8 * vs));
    const long long per = (nmcu + 255) / 256;
    const long long m0 = per * tid, m1 = m0 + per < nmcu ? m0 + per : nmcu;
    short* coef = coef_all + im.coef_off * 64;
    if (im.restart_interval > 0) {                      // predictions start at 0 in every restart interval: a thread per interval
        const long long ri = im.restart_interval;
        for (long long k = tid; k * ri < nmcu; k += 256) {
            int run[3] = {0, 0, 0};
            const long long e = (k + 1) * ri < nmcu ? (k + 1) * ri : nmcu;
            for (long long m = k * ri; m < e; ++m)
                for (int b = 0; b < bpm; ++b) {
                    const int c = b < hv ? 0 : b - hv + 1;
                    run[c] += coef[(m * bpm + b) * 64];
                    coef[(m * bpm + b) * 64] = (short)run[c];
                }
        }
        return;
    }
    int s[3] = {0, 0, 0};
    for (long long m = m0; m < m1; ++m)
        for (int b = 0; b < bpm; ++b) s[b < hv ? 0 : b - hv + 1] += coef[(m * bpm + b) * 64];
    for (int c = 0; c < 3; ++c) part[c][tid] = s[c];
    __syncthreads();
    for (int d = 1; d < 256; d <<= 1) {
        int a[3];
        for (int c = 0; c < 3; ++c) a[c] = tid >= d ? part[c][tid - d] : 0;
        __syncthreads();
        for (int c = 0; c < 3; ++c) part[c][tid] += a[c];
        __syncthreads();
    }
    int run[3];
    for (int c = 0; c < 3; ++c) run[c] = part[c][tid] - s[c];
    for (long long m = m0; m < m1; ++m)
        for (int b = 0; b < bpm; ++b) {
            const int c = b < hv ? 0 : b - hv + 1;
            run[c] += coef[(m * bpm + b) * 64];
            coef[(m * bpm + b) * 64] = (short)run[c];
        }
}

// jidctint.c jpeg_idct_islow: one dimension of the LL&M butterfly (CONST_BITS = 13), results descaled by `SHIFT`
template <int SHIFT>
__device__ __forceinline__ void jp_idct8(const int* v, int stride, int* o, int ostride) {
    int z2 = v[2 * stride], z3 = v[6 * stride];
    int z1 = (z2 + z3) * 4433;
    const int t2 = z1 - z3 * 15137, t3 = z1 + z2 * 6270;
    const int t0 = (v[0] + v[4 * stride]) << 13, t1 = (v[0] - v[4 * stride]) << 13;
    const int t10 = t0 + t3, t13 = t0 - t3, t11 = t1 + t2, t12 = t1 - t2;
    int o0 = v[7 * stride], o1 = v[5 * stride], o2 = v[3 * stride], o3 = v[stride];
    z1 = o0 + o3;
    z2 = o1 + o2;
    z3 = o0 + o2;
    int z4 = o1 + o3;
    const int z5 = (z3 + z4) * 9633;
    o0 *= 2446;
    o1 *= 16819;
    o2 *= 25172;
    o3 *= 12299;
    z1 *= -7373;
    z2 *= -20995;
    z3 = z3 * -16069 + z5;
    z4 = z4 * -3196 + z5;
    o0 += z1 + z3;
    o1 += z2 + z4;
    o2 += z2 + z3;
    o3 += z1 + z4;
    constexpr int R = 1 << (SHIFT - 1);
    o[0] = (t10 + o3 + R) >> SHIFT;
    o[7 * ostride] = (t10 - o3 + R) >> SHIFT;
    o[ostride] = (t11 + o2 + R) >> SHIFT;
    o[6 * ostride] = (t11 - o2 + R) >> SHIFT;
    o[2 * ostride] = (t12 + o1 + R) >> SHIFT;
    o[5 * ostride] = (t12 - o1 + R) >> SHIFT;
    o[3 * ostride] = (t13 + o0 + R) >> SHIFT;
    o[4 * ostride] = (t13 - o0 + R) >> SHIFT;
}

// planes of an image (bytes from planes + coef_off * 64): Y [my vs 8][mx hs 8], then Cb, Cr [my 8][mx 8]
__global__ void __launch_bounds__(128) jpeg_idct_kernel(const JpegImage* __restrict__ images, const short* __restrict__ coef_all,
                                                        unsigned char* __restrict__ planes) {
    const JpegImage& im = images[blockIdx.y];
    const int hs = im.ncomp == 1 ? 1 : im.hs, vs = im.ncomp == 1 ? 1 : im.vs;
    const int hv = hs * vs, bpm = im.ncomp == 1 ? 1 : hv + 2;
    const int mx = (im.width + 8 * hs - 1) / (8 * hs), my = (im.height + 8 * vs - 1) / (8 * vs);
    const long long b = (long long)blockIdx.x * 128 + threadIdx.x;
    if (b >= (long long)mx * my * bpm) return;
    const long long mcu = b / bpm;
    const int k = (int)(b - mcu * bpm);
    const int comp = k < hv ? 0 : k - hv + 1;
    const int mcx = (int)(mcu % mx), mcy = (int)(mcu / mx);
    int bx, by, pw;
    unsigned char* plane = planes + im.coef_off * 64;
    const long long ysize = (long long)mx * my * hv * 64;
    if (comp == 0) {
        bx = mcx * hs + k % hs;
        by = mcy * vs + k / hs;
        pw = mx * hs * 8;
    } else {
        bx = mcx;
        by = mcy;
        pw = mx * 8;
        plane += ysize + (long long)(comp - 1) * mx * my * 64;
    }
    const uint4* cp = reinterpret_cast<const uint4*>(coef_all + (im.coef_off + b) * 64);
    int x[64], ws[64];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const uint4 q = cp[i];
        const unsigned u[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            x[i * 8 + 2 * j] = (int)(short)(u[j] & 0xffff) * im.quant[comp][i * 8 + 2 * j];
            x[i * 8 + 2 * j + 1] = ((int)u[j] >> 16) * im.quant[comp][i * 8 + 2 * j + 1];
        }
    }
#pragma unroll
    for (int c = 0; c < 8; ++c) jp_idct8<11>(x + c, 8, ws + c, 8);       // pass 1: columns (CONST_BITS - PASS1_BITS)
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        int o[8];
        jp_idct8<18>(ws + r * 8, 1, o, 1);                                // pass 2: rows (CONST_BITS + PASS1_BITS + 3)
        unsigned lo = 0, hi = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            int a = o[j] + 128, c2 = o[4 + j] + 128;
            a = a < 0 ? 0 : (a > 255 ? 255 : a);
            c2 = c2 < 0 ? 0 : (c2 > 255 ? 255 : c2);
            lo |= (unsigned)a << (8 * j);
            hi |= (unsigned)c2 << (8 * j);
        }
        *reinterpret_cast<uint2*>(plane + (size_t)(by * 8 + r) * pw + bx * 8) = make_uint2(lo, hi);
    }
}

__global__ void __launch_bounds__(256) jpeg_color_kernel(const JpegImage* __restrict__ images, const unsigned char* __restrict__ planes,
                                                         unsigned char* __restrict__ out) {
    const JpegImage& im = images[blockIdx.y];
    const int W = im.width, H = im.height;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long long)W * H) return;
    const int y = (int)(idx / W), x = (int)(idx - (long long)y * W);
    const int hs = im.ncomp == 1 ? 1 : im.hs, vs = im.ncomp == 1 ? 1 : im.vs;
    const int mx = (W + 8 * hs - 1) / (8 * hs), my = (H + 8 * vs - 1) / (8 * vs);
    const unsigned char* py = planes + im.coef_off * 64;
    const int yw = mx * hs * 8;
    const int Y = py[(size_t)y * yw + x];
    unsigned char* o = out + im.out_off + ((size_t)y * W + x) * 3;
    if (im.ncomp == 1) {
        o[0] = o[1] = o[2] = (unsigned char)Y;
        return;
    }
    const int cw = mx * 8;
    const unsigned char* pc[2];
    pc[0] = py + (size_t)mx * my * hs * vs * 64;
    pc[1] = pc[0] + (size_t)mx * my * 64;
    int cc[2];
    if (hs == 1) {
        for (int c = 0; c < 2; ++c) cc[c] = pc[c][(size_t)y * cw + x];
    } else {
        const int dw = (W + 1) >> 1;
        const int cx = x >> 1;
        if (vs == 1) {                          // h2v1_fancy_upsample
            for (int c = 0; c < 2; ++c) {
                const unsigned char* r = pc[c] + (size_t)y * cw;
                const int p = r[cx];
                if (x & 1) cc[c] = cx == dw - 1 ? p : (3 * p + r[cx + 1] + 2) >> 2;
                else cc[c] = cx == 0 ? p : (3 * p + r[cx - 1] + 1) >> 2;
            }
        } else {                                // h2v2_fancy_upsample, context rows duplicated at the image's edges
            const int dh = (H + 1) >> 1;
            const int cy = y >> 1;
            const int oy = (y & 1) ? (cy + 1 < dh ? cy + 1 : dh - 1) : (cy > 0 ? cy - 1 : 0);
            for (int c = 0; c < 2; ++c) {
                const unsigned char* r0 = pc[c] + (size_t)cy * cw;
                const unsigned char* r1 = pc[c] + (size_t)oy * cw;
                const int s = 3 * r0[cx] + r1[cx];
                if (x & 1) cc[c] = cx == dw - 1 ? (4 * s + 7) >> 4 : (3 * s + 3 * r0[cx + 1] + r1[cx + 1] + 7) >> 4;
                else cc[c] = cx == 0 ? (4 * s + 8) >> 4 : (3 * s + 3 * r0[cx - 1] + r1[cx - 1] + 8) >> 4;
            }
        }
    }
    const int cb = cc[0] - 128, cr = cc[1] - 128;             // jdcolor.c build_ycc_rgb_table: FIX(1.402), FIX(0.34414), ...
    int r = Y + ((91881 * cr + 32768) >> 16);
    int g = Y + ((-22554 * cb + 32768 - 46802 * cr) >> 16);
    int b = Y + ((116130 * cb + 32768) >> 16);
    o[0] = (unsigned char)(r < 0 ? 0 : (r > 255 ? 255 : r));
    o[1] = (unsigned char)(g < 0 ? 0 : (g > 255 ? 255 : g));
    o[2] = (unsigned char)(b < 0 ? 0 : (b > 255 ? 255 : b));
}

}  // namespace
}  // namespace clipmi

using namespace clipmi;

extern "C" int64_t clipmi_jpeg_workspace_bytes(int64_t total_blocks, int ntables) {
    if (total_blocks < 0 || ntables < 0) return -1;
    return (int64_t)align_up((size_t)ntables * sizeof(JpLut), 256) + (int64_t)align_up((size_t)total_blocks * 128, 256) +
           (int64_t)align_up((size_t)total_blocks * 64, 256);
}

extern "C" int clipmi_jpeg_decode_rgb8(void* streams_dev, void* images_dev, int n, const void* tables_dev, int ntables,
                                       int64_t total_blocks, int64_t max_blocks, int64_t max_pixels, void* out_dev, int32_t* status_dev,
                                       void* ws_dev, int64_t ws_bytes, void* stream) {
    static_assert(sizeof(JpegImage) == sizeof(clipmi_jpeg_image) && sizeof(JpegImage) == 288, "clipmi_jpeg_image layout");
    if (n == 0) return 0;
    if (!streams_dev || !images_dev || !tables_dev || !out_dev || !status_dev || !ws_dev || n < 0 || ntables < 1 || total_blocks < 1 ||
        max_blocks < 1 || max_blocks > total_blocks || max_pixels < 1)
        return set_err(CLIPMI_EINVAL, "jpeg_decode_rgb8: bad arguments");
    if (ws_bytes < clipmi_jpeg_workspace_bytes(total_blocks, ntables))
        return set_err(CLIPMI_EINVAL, "jpeg_decode_rgb8: workspace of %lld bytes, %lld needed", (long long)ws_bytes,
                       (long long)clipmi_jpeg_workspace_bytes(total_blocks, ntables));
    if ((max_blocks + 127) / 128 > 0x7fffffffLL || (max_pixels + 255) / 256 > 0x7fffffffLL || n > 65535)
        return set_err(CLIPMI_EINVAL, "jpeg_decode_rgb8: batch too large for one launch");
    hipStream_t st = as_stream(stream);
    Arena ar(ws_dev, (size_t)ws_bytes);
    JpLut* luts = ar.take<JpLut>((size_t)ntables);
    short* coef = ar.take<short>((size_t)total_blocks * 64);
    unsigned char* planes = ar.take<unsigned char>((size_t)total_blocks * 64);
    JpegImage* images = static_cast<JpegImage*>(images_dev);
    hipLaunchKernelGGL(jpeg_unstuff_kernel, dim3((unsigned)n), dim3(256), 0, st, static_cast<unsigned char*>(streams_dev), images);
    CLIPMI_CHECK_LAUNCH("jpeg_unstuff_kernel");
    if (hipMemsetAsync(coef, 0, (size_t)total_blocks * 128, st) != hipSuccess) return set_err(CLIPMI_EHIP, "jpeg_decode_rgb8: memset");
    hipLaunchKernelGGL(jpeg_build_luts_kernel, dim3((unsigned)ntables), dim3(256), 0, st, static_cast<const unsigned char*>(tables_dev), luts);
    CLIPMI_CHECK_LAUNCH("jpeg_build_luts_kernel");
    hipLaunchKernelGGL(jpeg_huffman_kernel, dim3((unsigned)n), dim3(JP_T), 0, st, static_cast<const unsigned char*>(streams_dev), images,
                       luts, coef, status_dev);
    CLIPMI_CHECK_LAUNCH("jpeg_huffman_kernel");
    hipLaunchKernelGGL(jpeg_dc_kernel, dim3((unsigned)n), dim3(256), 0, st, images, coef);
    CLIPMI_CHECK_LAUNCH("jpeg_dc_kernel");
    hipLaunchKernelGGL(jpeg_idct_kernel, dim3((unsigned)((max_blocks + 127) / 128), (unsigned)n), dim3(128), 0, st, images, coef, planes);
    CLIPMI_CHECK_LAUNCH("jpeg_idct_kernel");
    hipLaunchKernelGGL(jpeg_color_kernel, dim3((unsigned)((max_pixels + 255) / 256), (unsigned)n), dim3(256), 0, st, images, planes,
                       static_cast<unsigned char*>(out_dev));
    CLIPMI_CHECK_LAUNCH("jpeg_color_kernel");
    return 0;
}
