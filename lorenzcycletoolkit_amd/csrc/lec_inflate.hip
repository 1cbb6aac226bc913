// lec_inflate.hip -- zlib / deflate streams inflated on the GPU, and the chunks of a NetCDF-4 variable put in place
// (include/lec_hip.h: lec_inflate, lec_chunk_scatter).
//
// Why: a deflated NetCDF-4 file (what the CDS delivers for ERA5) is bound by the HOST's inflate -- 3.75 GB/s of decoded data with
// 16 threads against a 57 GB/s link (profiles/r03_notes.md) -- and the reference's netCDF4 / HDF5 stack inflates in one thread.
// Here the link carries the compressed chunks as they lie in the file and the GPU inflates them: thousands of independent
// streams per batch of time steps (one per HDF5 chunk), ONE WAVE PER STREAM.
//
// A deflate stream is serial: the position of a code is known only when the code before it is decoded.  A wave breaks that
// chain by speculation: lane l decodes the token (literal, or length + distance with their extra bits, or end-of-block) that WOULD
// start at bit (base + l) -- 64 candidate positions, a 10-bit table lookup each in LDS (+ 9 bits for the distance code) --, then a
// short scalar walk follows the true chain through the lanes (position 0, then 0 + bits(0), ...), typically 6-9 tokens per round;
// output offsets are popcounts of the chain mask, and the wave writes: literals at once, matches one after the other with all 64
// lanes copying.  Codes longer than the lookup width (rare symbols) are resolved only when the walk actually lands on them
// (canonical decode, count / sorted-symbol arrays).
//
// Output goes through an 8 KiB ring in LDS (the recent history LZ77 matches mostly refer to) and is flushed to HBM in 16-byte
// pieces; a match that reaches further back than the ring reads the flushed bytes from HBM (a fence orders the wave's own earlier
// stores, taken lazily -- only when such a match occurs).  12.6 KiB of LDS and ~90 VGPRs per wave (89 / 92 in the two
// instantiations): 12 waves per CU, 3072 streams in flight; the 4 KiB-ring instantiation (flags bit 1) 8.5 KiB and 18 waves per CU.
// Measured (profiles/r03_notes.md 2b, ERA5-like 361 x 720 int16 chunks, shuffle + deflate 4, >= 6000 streams in flight): 40-44 GB/s of
// decoded data on noisy fields (ratio 1.3), 119-126 GB/s on smooth ones (ratio 1.8); the kernel is bound by instruction issue
// (10.8 scalar + 8.1 vector instructions per decoded byte on noisy fields; half of a round is its ~3 matches), not by memory.
//
// Every loop is bounded by the stream's bit length / the output size, and every descriptor is checked on the device before its
// stream is touched (source AND destination ranges against the sizes of the two allocations, ABI 8): malformed input ends with a
// status code, never a hang or an out-of-bounds access.
// The kernel contains no FLAT memory instruction (tests/test_build_cpu.py checks the ISA): LDS traffic is DS, HBM traffic is
// GLOBAL.  A FLAT access that resolves to LDS is not ordered with the wave's DS operations by wave_sync() (a wavefront-scope fence
// emits no s_waitcnt), which is how round 3's timing builds came to read stale ring bytes in the far-match form.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/lec_hip.h"
#include "lec_internal.h"

// measurement knobs (defaults = what ships)
#ifndef LEC_INFLATE_LITBITS
#define LEC_INFLATE_LITBITS 10
#endif
#ifndef LEC_INFLATE_DISTBITS
#define LEC_INFLATE_DISTBITS 9
#endif
#ifndef LEC_INFLATE_RING
#define LEC_INFLATE_RING 8192
#endif
#ifndef LEC_INFLATE_TIMING
#define LEC_INFLATE_TIMING 0          // debug builds (tools/probes/inflate_timing.py): 1..4 = time the round's window / decode / walk / write
#endif                                // phase, 6..8 = within the write phase: offsets + literals / the matches / sync + flush; 9 = a census of far / all / overlapping matches
#define LEC_TICK(k, var) if (LEC_INFLATE_TIMING == (k)) var = (uint32_t)__builtin_amdgcn_s_memtime()
#ifndef LEC_INFLATE_C_WALK
#define LEC_INFLATE_C_WALK 0
#endif

namespace {

constexpr int kLitBits = LEC_INFLATE_LITBITS;        // lookup width of the literal / length code (codes up to 15 bits: the rest resolves on demand)
constexpr int kDistBits = LEC_INFLATE_DISTBITS;
constexpr int kRingDefault = LEC_INFLATE_RING;       // LDS history ring (bytes, power of two): what most matches refer to
#ifndef LEC_INFLATE_RING_SHORT
#define LEC_INFLATE_RING_SHORT 4096
#endif
constexpr int kRingShort = LEC_INFLATE_RING_SHORT;                     // ... for streams whose matches stay close (flags bit 1): 18 instead of 12 waves per CU

enum { T_LIT = 0, T_MATCH = 1, T_EOB = 2, T_SLOW = 3, T_BAD = 4 };

// the order in which a dynamic block header lists the lengths of the code-length code (RFC 1951, 3.2.7)
__constant__ uint8_t kClOrder[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

template <int kRing>
struct __attribute__((aligned(16))) InflateLds {
    uint8_t ring[kRing];
    uint16_t lit[1 << kLitBits];    // (symbol << 4) | code length; 0: not in the table (longer code, or no such code)
    uint16_t dist[1 << kDistBits];
    uint16_t lsym[288];             // symbols sorted by (code length, symbol): canonical decode of the long codes
    uint16_t dsym[32];
    uint32_t lcnt[16], dcnt[16];    // number of codes of each length
    uint16_t clt[128];              // the code-length code of a dynamic block header (<= 7 bits)
    uint8_t lens[320];              // code lengths as read from the header: 288 literal/length + 32 distance
};

__device__ __forceinline__ uint32_t rl(uint32_t v, uint32_t lane) { return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)lane); }
__device__ __forceinline__ uint32_t wl(uint32_t value, uint32_t lane, uint32_t old) {      // `old` with lane `lane` set to the uniform `value`
    return (uint32_t)__lane_id() == lane ? value : old;
}
// exclusive prefix sum over the 64 lanes in the vector ALU (DPP row shifts, then the row totals handed on): no LDS, no scalar loop
__device__ __forceinline__ uint32_t wave_exclusive_sum(uint32_t x) {
    int v = (int)x;
    v += __builtin_amdgcn_update_dpp(0, v, 0x111 /* row_shr:1 */, 0xF, 0xF, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x112 /* row_shr:2 */, 0xF, 0xF, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x114 /* row_shr:4 */, 0xF, 0xF, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x118 /* row_shr:8 */, 0xF, 0xF, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x142 /* row_bcast:15 */, 0xA, 0xF, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x143 /* row_bcast:31 */, 0xC, 0xF, false);
    return (uint32_t)v - x;
}

// One wave per workgroup: its LDS operations execute in program order, so only the COMPILER has to be kept from moving them
// across this point (an s_barrier would also wait for the prefetched input block, every round).
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// The compressed stream seen through 64-dword blocks held in registers (lane l: dword blk * 64 + l, the one 64 further, and a third
// block in flight): any dword the decoder needs is a v_readlane away.  The block being loaded is never read -- it becomes readable
// one block later, when the load has long completed --, so no round waits for memory.
struct BitIn {
    const uint32_t* in32;
    uint32_t nwords;                // dwords that may be read (the buffer's padding included)
    uint32_t blk;
    uint32_t a, b, c;               // blocks blk, blk + 1 (readable), blk + 2 (in flight)
    int lane;
    __device__ __forceinline__ uint32_t load(uint32_t block) const {
        const uint32_t i = block * 64u + (uint32_t)lane;
        return i < nwords ? in32[i] : 0u;
    }
    __device__ __forceinline__ void reset(uint32_t bitpos) { blk = (bitpos >> 5) >> 6; a = load(blk); b = load(blk + 1); c = load(blk + 2); }
    // afterwards dwords [bitpos / 32, bitpos / 32 + 64) are held
    __device__ __forceinline__ void seek(uint32_t bitpos) {
        const uint32_t rel = (bitpos >> 5) - blk * 64u;        // (wraps when bitpos lies before the held range)
        if (rel >= 64u) {
            if (rel < 128u) { a = b; b = c; ++blk; c = load(blk + 2); }
            else reset(bitpos);
        }
    }
    __device__ __forceinline__ uint32_t dword(uint32_t d) const {      // d wave-uniform, within the held range: no branch
        const uint32_t rel = d - blk * 64u;
        const uint32_t x = rl(a, rel & 63u), y = rl(b, rel & 63u);
        return rel < 64u ? x : y;
    }
    __device__ __forceinline__ uint32_t peek(uint32_t bitpos) const {  // 32 bits at a wave-uniform position
        const uint32_t d = bitpos >> 5, s = bitpos & 31u;
        const uint64_t w = ((uint64_t)dword(d + 1) << 32) | dword(d);
        return (uint32_t)(w >> s);
    }
};

__device__ __forceinline__ void length_of(uint32_t c, uint32_t& base, uint32_t& extra) {      // c = symbol - 257, 0..28
    if (c < 8u) { base = 3u + c; extra = 0u; }
    else if (c == 28u) { base = 258u; extra = 0u; }
    else { extra = (c >> 2) - 1u; base = 3u + ((4u + (c & 3u)) << extra); }
}
__device__ __forceinline__ void distance_of(uint32_t d, uint32_t& base, uint32_t& extra) {    // d = 0..29
    if (d < 4u) { base = 1u + d; extra = 0u; }
    else { extra = (d >> 1) - 1u; base = 1u + ((2u + (d & 1u)) << extra); }
}

// canonical decode of one code from the low bits of `w` (first stream bit = most significant code bit); -1: no such code
__device__ __forceinline__ int canon(uint64_t w, const uint32_t* cnt, const uint16_t* sorted, uint32_t& len_out) {
    int code = 0, first = 0, index = 0;
    for (int len = 1; len <= 15; ++len) {
        code |= (int)(w & 1u); w >>= 1;
        const int count = (int)cnt[len];
        if (code - count < first) { len_out = (uint32_t)len; return (int)sorted[index + (code - first)]; }
        index += count; first += count; first <<= 1; code <<= 1;
    }
    return -1;
}

// Lookup table + canonical arrays from `n` code lengths in LDS.  0: ok (incomplete codes are allowed: their gaps decode as
// "no such code"); 1: over-subscribed.
__device__ int build_code(const uint8_t* lens, int n, uint16_t* table, int bits, uint16_t* sorted, uint32_t* cnt, int lane) {
    int mylen[5];
#pragma unroll
    for (int g = 0; g < 5; ++g) { const int s = g * 64 + lane; mylen[g] = s < n ? (int)lens[s] : 0; }
    for (int i = lane; i < (1 << bits); i += 64) table[i] = 0;
    if (lane < 16) cnt[lane] = 0;
    wave_sync();
    uint32_t code = 0, index = 0;
    int left = 1;
    const uint64_t below = (1ull << lane) - 1ull;
    for (int len = 1; len <= 15; ++len) {
        uint64_t m[5];
        uint32_t c = 0;
#pragma unroll
        for (int g = 0; g < 5; ++g) { m[g] = __ballot(mylen[g] == len); c += (uint32_t)__popcll(m[g]); }
        left = (left << 1) - (int)c;
        if (left < 0) return 1;
        if (c == 0) { code <<= 1; continue; }
        if (lane == 0) cnt[len] = c;
        uint32_t run = 0;
#pragma unroll
        for (int g = 0; g < 5; ++g) {
            if (mylen[g] == len) {
                const uint32_t r = run + (uint32_t)__popcll(m[g] & below);
                const uint32_t sym = (uint32_t)(g * 64 + lane);
                sorted[index + r] = (uint16_t)sym;
                if (len <= bits) {
                    const uint32_t rev = __brev(code + r) >> (32 - len);
                    const uint16_t e = (uint16_t)((sym << 4) | (uint32_t)len);
                    for (uint32_t at = rev; at < (1u << bits); at += (1u << len)) table[at] = e;
                }
            }
            run += (uint32_t)__popcll(m[g]);
        }
        code = (code + c) << 1;
        index += c;
    }
    wave_sync();
    return 0;
}

struct InflateParams {
    const uint8_t* src; long long src_bytes;
    const long long* desc; int n; int flags;
    uint8_t* dst; long long dst_bytes; int* status;
};

// HDF5's Fletcher-32 (H5_checksum_fletcher32: 16-bit big-endian words, sums folded with end-around carry) of `len` bytes, by one wave.
// With S1 = sum w_i and S2 = sum (n - i) w_i exact in 64 bits, the folded sums are ((x - 1) mod 65535) + 1 for x > 0.
__device__ uint32_t fletcher32_wave(const uint8_t* p, long long len, int lane) {
    const long long n = (len + 1) / 2;                           // an odd last byte counts as (byte << 8)
    // S1 = sum w_i and S2 = sum (n - i) w_i modulo 65535, weights reduced and the partial sums folded every 65536 words per lane (terms
    // below 2^32: no 64-bit wrap-around however large the chunk, as hdf5_lite._fletcher32 folds on the host); a sum that is a
    // multiple of 65535 folds to 65535 unless every word was zero
    unsigned long long s1 = 0, s2 = 0;
    bool nz = false;
    long long it = 0;
    for (long long i = lane; i < n; i += 64, ++it) {
        const uint32_t hi = p[2 * i], lo = (2 * i + 1 < len) ? p[2 * i + 1] : 0u;
        const unsigned long long w = (hi << 8) | lo;
        nz = nz || (w != 0);
        s1 += w;
        s2 += (unsigned long long)((n - i) % 65535ll) * w;
        if ((it & 0xffff) == 0xffff) { s1 %= 65535ull; s2 %= 65535ull; }
    }
    nz = __ballot(nz) != 0;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        s1 += ((unsigned long long)__shfl_xor((unsigned)(s1 >> 32), off) << 32) | __shfl_xor((unsigned)s1, off);
        s2 += ((unsigned long long)__shfl_xor((unsigned)(s2 >> 32), off) << 32) | __shfl_xor((unsigned)s2, off);
    }
    const uint32_t r1 = (uint32_t)(s1 % 65535ull), r2 = (uint32_t)(s2 % 65535ull);
    const uint32_t f1 = nz ? (r1 ? r1 : 65535u) : 0u, f2 = nz ? (r2 ? r2 : 65535u) : 0u;
    return (f2 << 16) | f1;
}

enum {
    ST_OK = 0, ST_HEADER = 1, ST_BLOCK_TYPE = 2, ST_STORED = 3, ST_CODE_LENGTHS = 4, ST_OVERSUBSCRIBED = 5, ST_BAD_CODE = 6,
    ST_DISTANCE = 7, ST_INPUT_END = 8, ST_OUTPUT_FULL = 9, ST_SIZE = 10, ST_STALLED = 11, ST_CHECKSUM = 12, ST_ADLER = 13
};

#ifdef LEC_INFLATE_WAVES
#define LEC_INFLATE_OCC __attribute__((amdgpu_waves_per_eu(LEC_INFLATE_WAVES)))
#else
#define LEC_INFLATE_OCC
#endif
template <int kRing>
__global__ void __launch_bounds__(64) LEC_INFLATE_OCC lec_inflate_kernel(const InflateParams P) {
    constexpr int kCap = kRing / 4;             // most output bytes one round of tokens may produce (up to its last match)
    constexpr int kFlushAt = kRing / 8;         // pending bytes that trigger a flush of the ring to HBM
    __shared__ InflateLds<kRing> L;
    const int lane = (int)threadIdx.x;
    const int s = (int)blockIdx.x;
    const long long src_off = P.desc[4 * s + 0], src_len = P.desc[4 * s + 1], dst_off = P.desc[4 * s + 2], dst_len = P.desc[4 * s + 3];
    uint8_t* const out = P.dst + dst_off;
    {
        // bit positions and output positions are 32-bit here: a stream of up to 256 MiB, an output of up to 2 GiB (an HDF5 chunk is < 4 GiB
        // by format, a sensible one a few MiB); the stream must lie inside the source buffer and its output, at a multiple of 16,
        // inside the destination buffer (the flush stores 16 bytes at a time)
        const long long n = src_len < 0 ? -src_len : src_len;
        if (n >= (1ll << 28) || dst_len < 0 || dst_len >= (1ll << 31) || src_off < 0 || src_off + n + ((P.flags & 1) ? 4 : 0) > P.src_bytes ||
            dst_off < 0 || (dst_off & 15) || dst_off + dst_len > P.dst_bytes) {
            if (lane == 0) { P.status[4 * s + 0] = ST_SIZE; P.status[4 * s + 1] = 0; P.status[4 * s + 2] = 0; P.status[4 * s + 3] = 0; }
            return;
        }
    }
    if (P.flags & 1) {
        // HDF5 filter 3: the stored chunk ends with the Fletcher-32 of everything before it (little-endian; the library also accepts
        // the byte-swapped form that 1.6.2 wrote on little-endian hosts, H5Zfletcher32.c)
        const long long n = src_len < 0 ? -src_len : src_len;
        const uint8_t* q = P.src + src_off;
        const uint32_t c = fletcher32_wave(q, n, lane);
        const uint32_t stored = (uint32_t)q[n] | ((uint32_t)q[n + 1] << 8) | ((uint32_t)q[n + 2] << 16) | ((uint32_t)q[n + 3] << 24);
        const uint32_t swapped = ((c & 0x00ff00ffu) << 8) | ((c >> 8) & 0x00ff00ffu);
        if (stored != c && stored != swapped) {
            if (lane == 0) { P.status[4 * s + 0] = ST_CHECKSUM; P.status[4 * s + 1] = 0; P.status[4 * s + 2] = (int)stored; P.status[4 * s + 3] = (int)c; }
            return;
        }
    }
    if (src_len < 0) {
        // the chunk was stored as it is (HDF5 skips an optional filter that does not pay): a plain copy
        const long long n = -src_len;
        const uint8_t* const raw = P.src + src_off;
        if (n == dst_len) {
            if (((uintptr_t)raw & 15u) == 0) {
                for (long long q = 16ll * lane; q + 16 <= n; q += 16 * 64) *(uint4*)(out + q) = *(const uint4*)(raw + q);
                for (long long q = (n & ~15ll) + lane; q < n; q += 64) out[q] = raw[q];
            } else {
                for (long long q = lane; q < n; q += 64) out[q] = raw[q];
            }
        }
        if (lane == 0) { P.status[4 * s + 0] = n == dst_len ? ST_OK : ST_SIZE; P.status[4 * s + 1] = 0; P.status[4 * s + 2] = (int)n; P.status[4 * s + 3] = 0; }
        return;
    }
    // a stream may start at any byte (chunks lie in the file as HDF5 put them): the dword view starts at the aligned address
    // below it, and every bit position carries the offset
    const uint32_t lead = (uint32_t)(src_off & 3);
    const uint8_t* const inb = P.src + (src_off - lead);
    BitIn in;
    in.in32 = (const uint32_t*)inb; in.lane = lane;
    {
        const long long words = (P.src_bytes - (src_off - lead)) >> 2;       // readable dwords from the aligned start (a few past the stream are touched)
        in.nwords = (uint32_t)(words < (1ll << 27) ? words : (1ll << 27));
    }
    const uint32_t src_end = lead + (uint32_t)src_len;          // in bytes from inb
    const uint32_t src_bits = src_end * 8u;
    const uint32_t out_len = (uint32_t)dst_len;
    uint32_t bitpos = lead * 8u + 16u;           // after the zlib header
    uint32_t opos = 0, flushed = 0, fenced = 0;
    int status = ST_OK, block = 0;
    in.reset(lead * 8u);
    {
        const uint32_t h = in.peek(lead * 8u);
        const uint32_t cmf = h & 0xffu, flg = (h >> 8) & 0xffu;
        if (src_len < 6 || (cmf & 0x0fu) != 8u || (cmf >> 4) > 7u || ((cmf << 8) | flg) % 31u != 0u || (flg & 0x20u)) status = ST_HEADER;
    }

    // ring -> HBM, whole 16-byte pieces (all of it when `all`)
    // zlib's own check of the DATA rides on the flush: Adler-32 is s1 = 1 + sum b_i, s2 = N + sum (N - i) b_i (mod 65521, i from 0, N
    // bytes); every lane keeps its share of both sums for the bytes it flushes -- 16 vector instructions per KiB of output
    uint32_t ad1 = 0, ad2 = 0;                  // this lane's sums, each < 65521 * 4096
    auto adler_bytes = [&](uint32_t word, uint32_t pos, uint32_t nbytes) {      // `nbytes` low bytes of `word`, the first at output position pos
        const uint32_t w0 = (out_len - pos) % 65521u;                            // weight of the first byte
#pragma unroll
        for (uint32_t b = 0; b < 4u; ++b) {
            const uint32_t v = b < nbytes ? (word >> (8u * b)) & 0xffu : 0u;
            ad1 += v;
            ad2 += v * ((w0 + 65521u - b) % 65521u);
        }
        ad1 %= 65521u; ad2 %= 65521u;
    };
    auto flush = [&](bool all) {
        const uint32_t end16 = opos & ~15u;
        for (uint32_t q = flushed + 16u * (uint32_t)lane; q < end16; q += 16u * 64u) {
            const uint4 v = *(const uint4*)&L.ring[q & (kRing - 1)];
            *(uint4*)(out + q) = v;
            adler_bytes(v.x, q, 4); adler_bytes(v.y, q + 4, 4); adler_bytes(v.z, q + 8, 4); adler_bytes(v.w, q + 12, 4);
        }
        if (end16 > flushed) flushed = end16;
        if (all) {
            for (uint32_t q = flushed + (uint32_t)lane; q < opos; q += 64u) {
                const uint8_t v = L.ring[q & (kRing - 1)];
                out[q] = v;
                adler_bytes(v, q, 1);
            }
            flushed = opos;
        }
    };

    uint32_t tsum = 0, trounds = 0, tmatches = 0;      // (timing builds)
    bool last = false;
    while (status == ST_OK && !last) {
        // ------------------------------------------------------------------ block header (wave-uniform)
        in.seek(bitpos);
        if (bitpos + 3u > src_bits) { status = ST_INPUT_END; break; }
        uint32_t hdr = in.peek(bitpos);
        last = (hdr & 1u) != 0u;
        const uint32_t btype = (hdr >> 1) & 3u;
        bitpos += 3;
        ++block;
        if (btype == 3u) { status = ST_BLOCK_TYPE; break; }
        if (btype == 0u) {
            // stored: to the next byte, LEN, ~LEN, the bytes
            bitpos = (bitpos + 7u) & ~7u;
            in.seek(bitpos);
            if (bitpos + 32u > src_bits) { status = ST_INPUT_END; break; }
            const uint32_t ll = in.peek(bitpos);
            uint32_t len = ll & 0xffffu;
            if (len != ((~ll >> 16) & 0xffffu)) { status = ST_STORED; break; }
            bitpos += 32;
            uint32_t at = bitpos >> 3;
            if ((uint64_t)at + len > (uint64_t)src_end) { status = ST_INPUT_END; break; }
            if (len > out_len - opos) { status = ST_OUTPUT_FULL; break; }
            while (len) {
                const uint32_t n = len < (uint32_t)kCap ? len : (uint32_t)kCap;
                for (uint32_t k = (uint32_t)lane; k < n; k += 64u) L.ring[(opos + k) & (kRing - 1)] = inb[at + k];
                wave_sync();
                opos += n; at += n; len -= n;
                if (opos - flushed >= (uint32_t)kFlushAt) flush(false);
            }
            bitpos = at * 8u;
            continue;
        }
        // ------------------------------------------------------------------ the two codes of this block
        int nlit = 288, ndist = 32;
        if (btype == 1u) {
            for (int i = lane; i < 320; i += 64) L.lens[i] = (uint8_t)(i < 144 ? 8 : i < 256 ? 9 : i < 280 ? 7 : i < 288 ? 8 : 5);
        } else {
            in.seek(bitpos);
            if (bitpos + 14u > src_bits) { status = ST_INPUT_END; break; }
            hdr = in.peek(bitpos);
            nlit = (int)(hdr & 31u) + 257; ndist = (int)((hdr >> 5) & 31u) + 1;
            const int ncl = (int)((hdr >> 10) & 15u) + 4;
            bitpos += 14;
            if (nlit > 286 || ndist > 30) { status = ST_CODE_LENGTHS; break; }
            // code-length code: ncl x 3 bits in the order 16 17 18 0 8 7 9 6 10 5 11 4 12 3 13 2 14 1 15
            if (lane < 19) L.lens[lane] = 0;
            wave_sync();
            for (int i = 0; i < ncl; ++i) {
                in.seek(bitpos);
                const uint32_t v = in.peek(bitpos) & 7u;
                bitpos += 3;
                if (lane == 0) L.lens[kClOrder[i]] = (uint8_t)v;
            }
            if (bitpos > src_bits) { status = ST_INPUT_END; break; }
            wave_sync();
            if (build_code(L.lens, 19, L.clt, 7, L.dsym, L.dcnt, lane)) { status = ST_OVERSUBSCRIBED; break; }
            // the nlit + ndist code lengths, run-length coded with that code
            int have = 0, prev = 0;
            const int want = nlit + ndist;
            bool bad = false;
            while (have < want) {
                in.seek(bitpos);
                const uint32_t w = in.peek(bitpos);
                const uint32_t e = L.clt[w & 127u];
                if (e == 0u) { bad = true; break; }
                const uint32_t cl = e & 15u, sym = e >> 4;
                bitpos += cl;
                int rep = 1, val = (int)sym;
                if (sym == 16u) { if (have == 0) { bad = true; break; } rep = 3 + (int)((w >> cl) & 3u); bitpos += 2; val = prev; }
                else if (sym == 17u) { rep = 3 + (int)((w >> cl) & 7u); bitpos += 3; val = 0; }
                else if (sym == 18u) { rep = 11 + (int)((w >> cl) & 127u); bitpos += 7; val = 0; }
                if (have + rep > want || bitpos > src_bits) { bad = true; break; }
                // literal/length lengths at lens[0..nlit), distance lengths at lens[288..288 + ndist)
                for (int r = lane; r < rep; r += 64) { const int i = have + r; L.lens[i < nlit ? i : 288 + (i - nlit)] = (uint8_t)val; }
                have += rep; prev = val;
            }
            if (bad) { status = ST_CODE_LENGTHS; break; }
            wave_sync();
            for (int i = nlit + lane; i < 288; i += 64) L.lens[i] = 0;
            for (int i = 288 + ndist + lane; i < 320; i += 64) L.lens[i] = 0;
            wave_sync();
            if (L.lens[256] == 0) { status = ST_CODE_LENGTHS; break; }      // no end-of-block code
        }
        wave_sync();
        if (build_code(L.lens, 288, L.lit, kLitBits, L.lsym, L.lcnt, lane) || build_code(L.lens + 288, 32, L.dist, kDistBits, L.dsym, L.dcnt, lane)) {
            status = ST_OVERSUBSCRIBED; break;
        }

        // ------------------------------------------------------------------ the block's tokens, up to 64 candidate positions a round
        bool eob = false;
        uint32_t rounds = 0;
        while (!eob) {
            uint32_t tk0 = 0, tk1 = 0;
            LEC_TICK(1, tk0);
            if (++rounds > src_bits + 8u) { status = ST_STALLED; break; }
            in.seek(bitpos);
            // 64 bits of the stream from bit (bitpos + lane)
            uint64_t win;
            {
                const uint32_t d0 = bitpos >> 5;
                const uint32_t u0 = in.dword(d0), u1 = in.dword(d0 + 1), u2 = in.dword(d0 + 2), u3 = in.dword(d0 + 3), u4 = in.dword(d0 + 4);
                const uint32_t o = (bitpos & 31u) + (uint32_t)lane, wi = o >> 5, sh = o & 31u;
                const uint32_t x0 = wi == 0u ? u0 : wi == 1u ? u1 : u2;
                const uint32_t x1 = wi == 0u ? u1 : wi == 1u ? u2 : u3;
                const uint32_t x2 = wi == 0u ? u2 : wi == 1u ? u3 : u4;
                win = (((uint64_t)x1 << 32) | x0) >> sh;
                if (sh) win |= (uint64_t)x2 << (64u - sh);
            }
            LEC_TICK(1, tk1);
            LEC_TICK(2, tk0);
            // the token that would start here -- straight-line code: every lane evaluates the literal AND the match reading (the second
            // table lookup included: its index is masked, so garbage bits are harmless) and selects; with 64 speculative positions some
            // lane takes every path anyway, and divergent branches cost scalar instructions on a unit the whole CU shares
            uint32_t type, used, value, dist;                // value: the literal byte, or the match length
            {
                const uint32_t e = L.lit[(uint32_t)win & ((1u << kLitBits) - 1u)];
                const uint32_t l0 = e & 15u, sym = e >> 4;
                const uint32_t c = sym - 257u;                                   // length code 0..28 if this is one
                const bool small = c < 8u, top = c >= 28u;
                const uint32_t lextra = (small || top) ? 0u : (c >> 2) - 1u;
                const uint32_t lbase = small ? 3u + c : top ? 258u : 3u + ((4u + (c & 3u)) << lextra);
                const uint32_t length = lbase + ((uint32_t)(win >> l0) & ((1u << lextra) - 1u));
                const uint32_t l1 = l0 + lextra;
                const uint32_t de = L.dist[(uint32_t)(win >> l1) & ((1u << kDistBits) - 1u)];
                const uint32_t dl = de & 15u, dsym = de >> 4;
                const uint32_t dextra = dsym < 4u ? 0u : ((dsym >> 1) - 1u) & 15u;
                const uint32_t dbase = dsym < 4u ? 1u + dsym : 1u + ((2u + (dsym & 1u)) << dextra);
                const uint32_t l2 = l1 + dl;
                dist = dbase + ((uint32_t)(win >> l2) & ((1u << dextra) - 1u));
                const bool is_len = sym - 257u <= 28u;
                type = e == 0u ? T_SLOW : sym < 256u ? T_LIT : sym == 256u ? T_EOB : !is_len ? T_BAD : de == 0u ? T_SLOW : dsym > 29u ? T_BAD : T_MATCH;
                value = sym < 256u ? sym : length;
                used = type == T_MATCH ? l2 + dextra : l0;
            }
            LEC_TICK(2, tk1);
            LEC_TICK(3, tk0);
            // Follow the true chain through the lanes.  One word per lane carries what the walk needs of a token: bits used (6) |
            // type (3) | output bytes (9); a literal -- the common case -- is recognised by one compare and costs one readlane.
            auto pack = [](uint32_t ty, uint32_t nbits, uint32_t val) {
                return nbits | (ty << 6) | ((ty == T_LIT ? 1u : ty == T_MATCH ? val : 0u) << 9) | (ty == T_LIT ? 0x80000000u : 0u);      // sign bit: a literal
            };
            uint32_t info = pack(type, used, value);
            uint32_t pos = 0, extra_out = 0;            // extra_out: output bytes of the chain's matches beyond one each
            uint64_t chain = 0;                         // lanes whose token is on the chain and writes output
            const uint32_t win_lo = (uint32_t)win, win_hi = (uint32_t)(win >> 32);
            for (;;) {
                // a run of literals: the scalar unit is shared by the whole CU, so this inner loop is kept to a handful of
                // scalar instructions and touches no vector register
                uint32_t inf = 0;
#if LEC_INFLATE_C_WALK
                while (pos < 64u) {
                    inf = rl(info, pos);
                    if ((int)inf >= 0) break;                                     // not a literal
                    chain |= 1ull << pos;
                    pos += inf & 63u;
                }
#else
                // written out: the compiler's version of the loop above is 14 scalar instructions per literal (three branches, a
                // 64-bit shift + or for the chain bit); these are 8.  Matches are taken here too (label 3: their output bytes go to
                // extra_out) as long as the round stays well below its output cap -- a round of a real field holds about three, and
                // each trip through the general path below costs as much as several literals.  Leaves with pos >= 64, or with `inf` =
                // the token at pos that the general path has to look at (end of block, a long code, a bad one, a match near the cap).
                if (pos < 64u) {
                    uint32_t tmp;
#define LEC_WALK_STEP \
                        "v_readlane_b32 %[inf], %[info], %[pos]\n\t" \
                        "s_cmp_lt_i32 %[inf], 0\n\t" \
                        "s_cbranch_scc0 3f\n\t" \
                        "s_bitset1_b64 %[chain], %[pos]\n\t" \
                        "s_and_b32 %[tmp], %[inf], 63\n\t" \
                        "s_add_u32 %[pos], %[pos], %[tmp]\n\t" \
                        "s_cmp_lt_u32 %[pos], 64\n\t"
                    // four literals per trip: a TAKEN branch empties the wave's instruction buffer, the exits here fall through
                    asm volatile(
                        "1:\n\t"
                        LEC_WALK_STEP "s_cbranch_scc0 2f\n\t"
                        LEC_WALK_STEP "s_cbranch_scc0 2f\n\t"
                        LEC_WALK_STEP "s_cbranch_scc0 2f\n\t"
                        LEC_WALK_STEP "s_cbranch_scc1 1b\n\t"
                        "s_branch 2f\n\t"
                        "3:\n\t"                                              // not a literal: a match (type field = 1)?
                        "s_bfe_u32 %[tmp], %[inf], 0x30006\n\t"
                        "s_cmp_eq_u32 %[tmp], 1\n\t"
                        "s_cbranch_scc0 2f\n\t"
                        "s_lshr_b32 %[tmp], %[inf], 9\n\t"                    // its length
                        "s_add_u32 %[extra], %[extra], %[tmp]\n\t"
                        "s_sub_u32 %[extra], %[extra], 1\n\t"
                        "s_cmp_gt_u32 %[extra], %[limit]\n\t"
                        "s_cbranch_scc1 4f\n\t"
                        "s_bitset1_b64 %[chain], %[pos]\n\t"
                        "s_and_b32 %[tmp], %[inf], 63\n\t"
                        "s_add_u32 %[pos], %[pos], %[tmp]\n\t"
                        "s_cmp_lt_u32 %[pos], 64\n\t"
                        "s_cbranch_scc1 1b\n\t"
                        "s_branch 2f\n\t"
                        "4:\n\t"                                              // near the cap: undo, the general path decides
                        "s_sub_u32 %[extra], %[extra], %[tmp]\n\t"
                        "s_add_u32 %[extra], %[extra], 1\n\t"
                        "2:"
                        : [inf] "=&s"(inf), [pos] "+s"(pos), [chain] "+s"(chain), [tmp] "=&s"(tmp), [extra] "+s"(extra_out)
                        : [info] "v"(info), [limit] "s"((uint32_t)(kCap - 65))
                        : "scc");
                }
#endif
                if (pos >= 64u) break;
                const uint32_t ty = (inf >> 6) & 7u;
                if (ty == T_SLOW) {
                    // a code longer than the lookup width (or none at all): decode this one position canonically, then look again
                    uint64_t w = ((uint64_t)rl(win_hi, pos) << 32) | rl(win_lo, pos);
                    uint32_t n = 0, tot;
                    const int sym = canon(w, L.lcnt, L.lsym, n);
                    uint32_t nty = T_BAD, nval = 0, ndis = 0;
                    tot = n;
                    if (sym >= 0 && sym < 256) { nty = T_LIT; nval = (uint32_t)sym; }
                    else if (sym == 256) nty = T_EOB;
                    else if (sym > 256 && sym <= 285) {
                        uint32_t base, extra;
                        length_of((uint32_t)sym - 257u, base, extra);
                        nval = base + ((uint32_t)(w >> tot) & ((1u << extra) - 1u));
                        tot += extra;
                        uint32_t dn = 0;
                        const int ds = canon(w >> tot, L.dcnt, L.dsym, dn);
                        if (ds >= 0 && ds <= 29) {
                            tot += dn;
                            distance_of((uint32_t)ds, base, extra);
                            ndis = base + ((uint32_t)(w >> tot) & ((1u << extra) - 1u));
                            tot += extra;
                            nty = T_MATCH;
                        }
                    }
                    type = wl(nty, pos, type); value = wl(nval, pos, value); dist = wl(ndis, pos, dist);
                    info = wl(pack(nty, tot & 63u, nval), pos, info);
                    continue;
                }
                if (ty == T_BAD) { status = ST_BAD_CODE; break; }
                if (ty == T_EOB) { pos += inf & 63u; eob = true; break; }
                const uint32_t n_out = inf >> 9;                                  // a match
                if ((uint32_t)__popcll(chain) + extra_out + n_out > (uint32_t)kCap) break;       // the next round starts at this token
                chain |= 1ull << pos;
                extra_out += n_out - 1u;
                pos += inf & 63u;
            }
            LEC_TICK(3, tk1);
            LEC_TICK(4, tk0);
            if (status != ST_OK) break;
            if (bitpos + pos > src_bits) { status = ST_INPUT_END; break; }
            const uint32_t produced = (uint32_t)__popcll(chain) + extra_out;
            if (produced > out_len - opos) { status = ST_OUTPUT_FULL; break; }
            const bool mine = (chain >> lane) & 1ull;
            uint64_t mm = chain & __ballot(type == T_MATCH);
            LEC_TICK(6, tk0);
            if (LEC_INFLATE_TIMING) tmatches += (uint32_t)__popcll(mm);
            // where each token's output starts: the output bytes of the chain tokens below it (a prefix sum in the vector ALU)
            const uint32_t ooff = wave_exclusive_sum(mine ? (type == T_MATCH ? value : type == T_LIT ? 1u : 0u) : 0u);
            const uint32_t mdesc = value | (dist << 9);                        // a match in one word (length <= 258, distance <= 32768)
            // literals
            if (mine && type == T_LIT) L.ring[(opos + ooff) & (kRing - 1)] = (uint8_t)value;
            // matches, in stream order, 64 bytes at a time
            const int safe_lo = (int)opos + kCap + 64 - kRing;                 // positions from here on are in the ring for the whole round (a round
                                                                                // writes at most kCap bytes up to its last match, then < 64 literals)
            LEC_TICK(6, tk1);
            LEC_TICK(7, tk0);
            uint32_t nx_md = 0, nx_off = 0;
            if (mm) { const uint32_t i = (uint32_t)__builtin_ctzll(mm); nx_md = rl(mdesc, i); nx_off = rl(ooff, i); }
            while (mm) {
                mm &= mm - 1ull;
                const uint32_t len = nx_md & 511u, d = nx_md >> 9, p = opos + nx_off;
                if (mm) {                                                       // the next match's two words: fetched before this one's copy
                    const uint32_t i = (uint32_t)__builtin_ctzll(mm);
                    nx_md = rl(mdesc, i); nx_off = rl(ooff, i);
                }
                if (d > p) { status = ST_DISTANCE; break; }
                const int from = (int)(p - d);
                if (LEC_INFLATE_TIMING == 9) { tsum += (from < safe_lo); trounds += (d < len); }     // (a census: far / overlapping matches)
                if (from < safe_lo && (uint32_t)from + (len < d ? len : d) > fenced) {
                    // the source was flushed by this wave's own earlier stores: make them visible to its loads
                    __threadfence();
                    fenced = flushed;
                }
                // three forms, chosen per match (wave-uniform): the common one -- source in the ring, no overlap -- is a bare LDS
                // read and write; an overlapping match (distance < length) repeats its d bytes; a far one reads the flushed bytes
                if (from >= safe_lo && d >= len) {
                    for (uint32_t k = (uint32_t)lane; k < len; k += 64u)
                        L.ring[(p + k) & (kRing - 1)] = L.ring[((uint32_t)from + k) & (kRing - 1)];
                } else if (from >= safe_lo) {
                    const float rd = __builtin_amdgcn_rcpf((float)d);          // (within an ulp: the remainder below is corrected by one step)
                    for (uint32_t k = (uint32_t)lane; k < len; k += 64u) {
                        const uint32_t q = (uint32_t)((float)k * rd);
                        int r = (int)k - (int)(q * d);
                        if (r < 0) r += (int)d; else if (r >= (int)d) r -= (int)d;
                        L.ring[(p + k) & (kRing - 1)] = L.ring[((uint32_t)from + (uint32_t)r) & (kRing - 1)];
                    }
                } else {
                    // (a source behind the ring lies more than kRing - kCap - 64 > 258 bytes back: it cannot overlap its output)
                    // BOTH loads are unconditional and the VALUES are selected: a conditional load lets the compiler select the POINTER
                    // (LDS or global) and read through one FLAT load -- which, when it resolves to LDS, is not ordered with the DS
                    // writes that filled the ring a moment ago (wave_sync() emits no s_waitcnt).  The global address is clamped into
                    // the flushed part; what the other load reads where it is not selected is in bounds and ignored.
                    const int glast = safe_lo > 0 ? safe_lo - 1 : 0;
                    for (uint32_t k = (uint32_t)lane; k < len; k += 64u) {
                        const int sp = from + (int)k;
                        const uint8_t ring_byte = L.ring[(uint32_t)sp & (kRing - 1)];
                        const uint8_t far_byte = __builtin_nontemporal_load(out + (sp < glast ? sp : glast));
                        L.ring[(p + k) & (kRing - 1)] = sp < safe_lo ? far_byte : ring_byte;
                    }
                }
                wave_sync();
            }
            LEC_TICK(7, tk1);
            LEC_TICK(8, tk0);
            if (status != ST_OK) break;
            wave_sync();
            opos += produced;
            bitpos += pos;
            if (opos - flushed >= (uint32_t)kFlushAt) flush(false);
            LEC_TICK(4, tk1);
            LEC_TICK(8, tk1);
            if (LEC_INFLATE_TIMING && LEC_INFLATE_TIMING != 9) { tsum += tk1 - tk0; ++trounds; }
        }
    }
    if (status == ST_OK) {
        flush(true);
        if (opos != out_len) status = ST_SIZE;
    }
    if (status == ST_OK) {
        // the stream ends with the Adler-32 of the data, big-endian, at the next byte boundary
        uint32_t s1 = ad1, s2 = ad2;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) { s1 += __shfl_xor(s1, off); s2 += __shfl_xor(s2, off); }      // 64 x 65521 fits
        s1 = (s1 + 1u) % 65521u;
        s2 = (s2 + out_len % 65521u) % 65521u;
        const uint32_t at = (bitpos + 7u) >> 3;
        if (at + 4u > src_end) status = ST_INPUT_END;
        else {
            const uint32_t stored = ((uint32_t)inb[at] << 24) | ((uint32_t)inb[at + 1] << 16) | ((uint32_t)inb[at + 2] << 8) | (uint32_t)inb[at + 3];
            if (stored != ((s2 << 16) | s1)) status = ST_ADLER;
        }
    }
    if (lane == 0) {
        P.status[4 * s + 0] = status; P.status[4 * s + 1] = block; P.status[4 * s + 2] = (int)opos; P.status[4 * s + 3] = (int)bitpos;
        if (LEC_INFLATE_TIMING) { P.status[4 * s + 1] = (int)tsum; P.status[4 * s + 2] = (int)tmatches; P.status[4 * s + 3] = (int)trounds; }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// chunks -> the contiguous raw sub-cube lec_ingest reads (un-shuffle, chunk tiling, edge chunks, level / time selection)
// ---------------------------------------------------------------------------------------------------------------------
struct ScatterParams {
    const uint8_t* src; long long src_bytes; const long long* chunk; int n_chunks, es, shuffled;
    int ct, ck, cj, ci;
    int t_base, n_tmap; const int* tmap; int n_kmap; const int* kmap; int j0;
    int nt, nl, ny, nx;
    uint8_t* out;
};

template <int ES>
__global__ void __launch_bounds__(256) lec_chunk_scatter_kernel(const ScatterParams p) {
    const int rows = p.ct * p.ck * p.cj;
    const int c = (int)(blockIdx.x / (unsigned)rows), row = (int)(blockIdx.x % (unsigned)rows);
    const int cj_ = row % p.cj, ck_ = (row / p.cj) % p.ck, ct_ = row / (p.cj * p.ck);
    const long long* ch = p.chunk + 5ll * c;
    const int ft = (int)ch[1] + ct_, fk = (int)ch[2] + ck_, fj = (int)ch[3] + cj_, fi0 = (int)ch[4];
    const int tt = ft - p.t_base;
    if (tt < 0 || tt >= p.n_tmap || fk < 0 || fk >= p.n_kmap) return;
    const int ot = p.tmap[tt], ok = p.kmap[fk], oj = fj - p.j0;
    if (ot < 0 || ot >= p.nt || ok < 0 || ok >= p.nl || oj < 0 || oj >= p.ny) return;
    const long long n_elem = (long long)rows * p.ci;
    if (ch[0] < 0 || ch[0] + n_elem * ES > p.src_bytes) return;      // a payload that does not lie inside src: skipped (the caller's rows keep what they held)
    const uint8_t* base = p.src + ch[0];
    const long long e0 = (long long)row * p.ci;
    uint8_t* orow = p.out + ((((long long)ot * p.nl + ok) * p.ny + oj) * p.nx) * ES;
    for (int i = (int)threadIdx.x; i < p.ci; i += (int)blockDim.x) {
        const int oi = fi0 + i;
        if (oi < 0 || oi >= p.nx) continue;                 // edge chunks are padded to the full chunk shape
        uint8_t b[ES];
        if (p.shuffled) {
#pragma unroll
            for (int q = 0; q < ES; ++q) b[q] = base[q * n_elem + e0 + i];
        } else {
#pragma unroll
            for (int q = 0; q < ES; ++q) b[q] = base[(e0 + i) * ES + q];
        }
        if constexpr (ES == 1) orow[oi] = b[0];
        else if constexpr (ES == 2) *(uint16_t*)(orow + 2ll * oi) = (uint16_t)(b[0] | (b[1] << 8));
        else if constexpr (ES == 4) *(uint32_t*)(orow + 4ll * oi) = (uint32_t)b[0] | ((uint32_t)b[1] << 8) | ((uint32_t)b[2] << 16) | ((uint32_t)b[3] << 24);
        else {
            uint64_t v = 0;
#pragma unroll
            for (int q = 0; q < ES; ++q) v |= (uint64_t)b[q] << (8 * q);
            *(uint64_t*)(orow + 8ll * oi) = v;
        }
    }
}

const char* status_text(int code) {
    switch (code) {
        case ST_HEADER: return "not a zlib stream (header)";
        case ST_BLOCK_TYPE: return "reserved deflate block type";
        case ST_STORED: return "stored block: length check failed";
        case ST_CODE_LENGTHS: return "dynamic block: bad code lengths";
        case ST_OVERSUBSCRIBED: return "over-subscribed Huffman code";
        case ST_BAD_CODE: return "invalid code in the stream";
        case ST_DISTANCE: return "match distance reaches before the start of the output";
        case ST_INPUT_END: return "compressed data end before the stream does";
        case ST_OUTPUT_FULL: return "stream holds more data than the chunk's size";
        case ST_SIZE: return "stream holds less data than the chunk's size";
        case ST_STALLED: return "decoder made no progress";
        case ST_CHECKSUM: return "fletcher32 checksum mismatch: the chunk is corrupt";
        case ST_ADLER: return "adler32 of the inflated data does not match the stream's trailer: the chunk is corrupt";
        default: return "unknown";
    }
}

}  // namespace

extern "C" int lec_inflate(const lec_inflate_args* a) {
    if (!a) return lec_set_error(LEC_ERR_ARG, "lec_inflate: null args");
    if (!a->src_d || !a->desc_d || !a->dst_d || !a->status_d) return lec_set_error(LEC_ERR_ARG, "lec_inflate: null pointer argument");
    if (a->n_streams < 1 || a->src_bytes < 8 || a->dst_bytes < 1) return lec_set_error(LEC_ERR_ARG, "lec_inflate: n_streams >= 1, src_bytes >= 8 and dst_bytes >= 1 needed");
    if (a->flags & ~3) return lec_set_error(LEC_ERR_ARG, "lec_inflate: unknown bits in flags");
    if (((uintptr_t)a->src_d & 3u) || ((uintptr_t)a->dst_d & 15u)) return lec_set_error(LEC_ERR_ARG, "lec_inflate: src_d must be 4-byte, dst_d 16-byte aligned");
    InflateParams p;
    p.src = (const uint8_t*)a->src_d; p.src_bytes = a->src_bytes; p.desc = (const long long*)a->desc_d; p.n = a->n_streams; p.flags = a->flags;
    p.dst = (uint8_t*)a->dst_d; p.dst_bytes = a->dst_bytes; p.status = a->status_d;
    if (a->flags & 2) hipLaunchKernelGGL(lec_inflate_kernel<kRingShort>, dim3((unsigned)a->n_streams), dim3(64), 0, (hipStream_t)a->stream, p);
    else hipLaunchKernelGGL(lec_inflate_kernel<kRingDefault>, dim3((unsigned)a->n_streams), dim3(64), 0, (hipStream_t)a->stream, p);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return lec_set_error(LEC_ERR_LAUNCH, hipGetErrorString(e));
    return LEC_OK;
}

extern "C" const char* lec_inflate_status_text(int code) { return code == 0 ? "ok" : status_text(code); }

extern "C" int lec_chunk_scatter(const lec_chunk_scatter_args* a) {
    if (!a) return lec_set_error(LEC_ERR_ARG, "lec_chunk_scatter: null args");
    if (!a->src_d || !a->chunk_d || !a->tmap_d || !a->kmap_d || !a->out_d) return lec_set_error(LEC_ERR_ARG, "lec_chunk_scatter: null pointer argument");
    if (a->n_chunks < 1 || a->ct < 1 || a->ck < 1 || a->cj < 1 || a->ci < 1 || a->nt < 1 || a->nl < 1 || a->ny < 1 || a->nx < 1 || a->n_tmap < 1 || a->n_kmap < 1)
        return lec_set_error(LEC_ERR_ARG, "lec_chunk_scatter: extents must be >= 1");
    if (a->elem_size != 1 && a->elem_size != 2 && a->elem_size != 4 && a->elem_size != 8) return lec_set_error(LEC_ERR_ARG, "lec_chunk_scatter: elem_size must be 1, 2, 4 or 8");
    if (a->src_bytes < (long long)a->ct * a->ck * a->cj * a->ci * a->elem_size) return lec_set_error(LEC_ERR_ARG, "lec_chunk_scatter: src_bytes is smaller than one chunk");
    const long long rows = (long long)a->ct * a->ck * a->cj;
    const long long blocks = rows * a->n_chunks;
    if (blocks > 0x7fffffffLL) return lec_set_error(LEC_ERR_UNSUPPORTED, "lec_chunk_scatter: more than 2^31-1 chunk rows in one call");
    ScatterParams p;
    p.src = (const uint8_t*)a->src_d; p.src_bytes = a->src_bytes; p.chunk = (const long long*)a->chunk_d; p.n_chunks = a->n_chunks; p.es = a->elem_size; p.shuffled = a->shuffled;
    p.ct = a->ct; p.ck = a->ck; p.cj = a->cj; p.ci = a->ci;
    p.t_base = a->t_base; p.n_tmap = a->n_tmap; p.tmap = a->tmap_d; p.n_kmap = a->n_kmap; p.kmap = a->kmap_d; p.j0 = a->j0;
    p.nt = a->nt; p.nl = a->nl; p.ny = a->ny; p.nx = a->nx; p.out = (uint8_t*)a->out_d;
    hipStream_t st = (hipStream_t)a->stream;
    const dim3 grid((unsigned)blocks), blk(256);
    if (a->elem_size == 1) hipLaunchKernelGGL(lec_chunk_scatter_kernel<1>, grid, blk, 0, st, p);
    else if (a->elem_size == 2) hipLaunchKernelGGL(lec_chunk_scatter_kernel<2>, grid, blk, 0, st, p);
    else if (a->elem_size == 4) hipLaunchKernelGGL(lec_chunk_scatter_kernel<4>, grid, blk, 0, st, p);
    else hipLaunchKernelGGL(lec_chunk_scatter_kernel<8>, grid, blk, 0, st, p);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return lec_set_error(LEC_ERR_LAUNCH, hipGetErrorString(e));
    return LEC_OK;
}
