// CLAHE and image histograms.
// Replaces cv2.createCLAHE(clipLimit, (t, t)).apply() (reference solex_util.py:532-533,
// clahe_apply.py:247) and feeds np.percentile / np.max (solex_util.py:535-537).
// Restated from OpenCV 4.x imgproc/clahe.cpp: per-tile histogram -> clip ->
// redistribute -> cumulative LUT (float32 scale, round half even) -> per-pixel
// bilinear blend of the four neighbouring tile LUTs in float32, unfused.
//
// The image is a few MB, so the stage is bound by the 65536-bin histogram scatter
// and by launch latency, not by HBM streaming.  Histogram: every workgroup owns a
// private 65536 x u16 LDS histogram (128 KiB) for a slice of <= 65535 pixels of one
// tile and flushes only its non-zero bins with global atomics.
#include <stdlib.h>
#include <type_traits>
#include <algorithm>
#include "shg_common.h"

namespace {
// A launch over several disks (blockIdx.z): every disk has a workspace of its own, `zs` bytes after the previous one, so a
// pointer into the first disk's workspace becomes the disk's by adding blockIdx.z * zs bytes.
template <typename T>
__device__ __forceinline__ T* zdisk(T* p, size_t zs, uint32_t z) {
    return reinterpret_cast<T*>(reinterpret_cast<char*>(const_cast<typename std::remove_const<T>::type*>(p)) + (size_t)z * zs);
}

constexpr int HIST16 = 65536;
// Workgroups flush their select histograms into one of SEL_SLOTS copies (blockIdx % SEL_SLOTS): ~500 workgroups adding to
// the same 256 addresses queue up behind each other in the memory-side atomic units; readers add the copies up.
constexpr int SEL_SLOTS = 8;
// The window (contrast stage only).  np.percentile(cl1, 10) needs the low byte's histogram of the pixels whose high byte holds the
// rank -- a second pass over the image, because which high byte that is is only known after the first.  But it hardly moves from one
// scan to the next: the blend kernel, which counts the high bytes anyway, also counts the low bytes of the SEL_NW high bytes around
// last scan's answer (kept in the workspace: SelWin), and np.max(cl1) is an atomic maximum.  When the ranks fall inside the window
// the second pass returns at once (k_select16_pass: every workgroup sees it after replaying the first pass), otherwise it runs as
// before; the result is the same either way.  Layout behind the SEL_SLOTS slot histograms of the 3-rank select:
// [SEL_WIN_SLOTS][SEL_NW][256] counts, then SelWin.
constexpr int SEL_NW = 8;
struct SelWin { uint32_t w0_cur, w0_next, max, pad; };     // first high byte of the window now / for the next scan; largest pixel
constexpr int SEL_WIN_SLOTS = 8;                           // (32 copies: the blend kernel no faster, the final pick 8 us slower)
constexpr int SEL_WIN_WORDS = SEL_WIN_SLOTS * SEL_NW * 256;
__device__ __forceinline__ uint32_t sel_window_origin(const SelWin* st) {        // (whatever the workspace held at first: a valid window)
    const uint32_t w0 = __hip_atomic_load(&st->w0_cur, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return w0 > 256u - SEL_NW ? 256u - SEL_NW : w0;
}
constexpr int SLICE_PX = 32768;        // pixels per workgroup (< 65536 so that u16 counters cannot wrap)
constexpr int kFusedMaxSliceRows = 2048;   // k_tile_hist16_slices<true> keeps a slice's row factors in LDS behind the histogram

// pixel (ty, tx, i) -> source coordinates with the bottom/right REFLECT_101 extension
template <typename T>
__device__ __forceinline__ uint32_t ext_pixel(const T* img, int64_t h, int64_t w, int64_t pitch, int64_t y, int64_t x) {
    if (y >= h) y = shg::reflect101(y, h);
    if (x >= w) x = shg::reflect101(x, w);
    return img[y * pitch + x];
}

// grid: (slices_per_tile, tiles*tiles).  16-bit images: LDS-private u16 histogram.
__global__ __launch_bounds__(1024) void k_tile_hist16(const uint16_t* __restrict__ img, int64_t h, int64_t w, int64_t pitch,
                                                      int tiles, int64_t th, int64_t tw, uint32_t* __restrict__ hist) {
    extern __shared__ uint32_t lh[];   // HIST16/2 dwords, two u16 counters each
    const int tile = blockIdx.y;
    const int64_t ty = tile / tiles, tx = tile % tiles;
    const int64_t area = th * tw;
    const int64_t p0 = (int64_t)blockIdx.x * SLICE_PX;
    const int64_t p1 = p0 + SLICE_PX < area ? p0 + SLICE_PX : area;
    for (int i = threadIdx.x; i < HIST16 / 2; i += 1024) lh[i] = 0;
    __syncthreads();
    {
        // the slice [p0, p1) of the tile's raster order, walked row by row: one wave per row, lanes along it.  (A division
        // per pixel to turn p into (yy, xx) made this kernel ALU bound: 26 us, see profiles/.)
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        const int twi = (int)tw;
        const int ya = (int)(p0 / tw), yb = (int)((p1 - 1) / tw);
        const int xs_first = (int)(p0 - (int64_t)ya * tw), xe_last = (int)(p1 - (int64_t)yb * tw);
        for (int yy = ya + wave; yy <= yb; yy += 16) {
            const int xs = yy == ya ? xs_first : 0, xe = yy == yb ? xe_last : twi;
            int64_t y = ty * th + yy;
            if (y >= h) y = shg::reflect101(y, h);
            const uint16_t* row = img + y * pitch;
            const int64_t xbase = tx * tw;
            for (int x0 = xs; x0 < xe; x0 += 64) {
                const int xx = x0 + lane;
                const bool in = xx < xe;
                int64_t x = xbase + (in ? xx : xs);
                if (x >= w) x = shg::reflect101(x, w);
                const uint32_t v = row[x];
                // the sky around the disc puts whole waves into one bin: its lanes are counted by one atomic
                const unsigned long long act = __ballot(in);
                const uint32_t v0 = __shfl(v, __ffsll((long long)act) - 1);
                const unsigned long long same = __ballot(in && v == v0);
                if (lane == __ffsll((long long)same) - 1) atomicAdd(&lh[v0 >> 1], (uint32_t)__popcll(same) << (16 * (v0 & 1)));
                if (in && v != v0) atomicAdd(&lh[v >> 1], (v & 1) ? 0x10000u : 1u);
            }
        }
    }
    __syncthreads();
    uint32_t* gh = hist + (int64_t)tile * HIST16;
    for (int i = threadIdx.x; i < HIST16 / 2; i += 1024) {
        const uint32_t c = lh[i];
        if (c & 0xffffu) atomicAdd(&gh[2 * i], c & 0xffffu);
        if (c >> 16) atomicAdd(&gh[2 * i + 1], c >> 16);
    }
}

// ---- the same histograms without global atomics ----------------------------------------------------------------------
// k_tile_hist16 flushes each slice's non-zero bins with device-scope atomics: ~20 000 per workgroup, 2.5 M per image,
// as many as there are pixels (26 us).  With room for them (shg_clahe_workspace_bytes_for) the slices' private u16
// histograms are stored as they are, coalesced (k_tile_hist16_slices), and one pass adds them up -- and, since it
// reads every bin anyway, also leaves what the next steps need: the 64-bin chunk sums the percentiles start from
// (k_hist_ranks) and, per 2048 bins, the clipped total and the clipped-off excess, so that the LUT no longer has to be
// built by one workgroup per tile (k_tile_lut16_blocks: 32 workgroups per tile instead of one).
// FUSED: the image whose histograms are taken does not exist yet -- it is made here, on the way, from the frame before it in
// single_image_process (Solex_recon.py:149-171): the circularised frame times its row factor (correct_transversalium2's last
// line, solex_util.py:515-516: saturate, truncate) through the crop / pad block (new[:, dx0:dx0+n] = img[:, sx0:sx0+n], the rest
// img[0, 0]).  The workgroup stores the pixels it has just formed into imgs (the "uncontrasted" image) and counts them: one
// read of the frame instead of k_scale_rows8's read + write, k_crop_pad's read + write and this kernel's read, and two launches
// less per scan.  (A grid that does not divide the image: the pixels of OpenCV's reflected border are formed a second time, for the
// count alone.)
struct FusedSrc {
    shg::PtrBatch raw;               // the frames before the row scaling, [h][raw_pitch]
    int64_t raw_pitch;
    const double* c;                 // [disk][h] row factors, or NULL (no transversalium correction: a plain crop / copy)
    int64_t sx0, dx0, ncopy;
};

__device__ __forceinline__ uint32_t scale_px(uint32_t px, double cy) {        // k_scale_rows: img * c[y], saturate, truncate
    double v = (double)px * cy;
    v = v > 65535.0 ? 65535.0 : v;
    return (uint32_t)(int)v;
}

// The same value with one fused multiply-add instead of a conversion and a multiplication: D = 2^52 + px is exact (the integer sits in
// the low mantissa bits), D * cy - 2^52 * cy is px * cy exactly, and the fma rounds it once -- as fl((double)px * cy) does.  kc = 2^52 * cy
// must be finite (rows_factor_for_fma); the conversion to int saturates, so the clamp is an integer minimum.  (The kernel's time was
// its VALU work: 23 instructions a pixel, the two conversions at a quarter of the rate.)
__device__ __forceinline__ uint32_t scale_px_fma(uint32_t px, double cy, double kc) {
    const double d = __hiloint2double(0x43300000, (int)px);
    const int q = __double2int_rz(__builtin_fma(d, cy, -kc));          // NaN -> 0, out of range -> INT_MAX / INT_MIN
    return (uint32_t)min(q, 65535);
}
// A row factor the fma form cannot take (NaN, or so large that 2^52 * c overflows) replaced by one that gives the same pixels:
// NaN -> every pixel (int)NaN = 0 = px * 0; huge -> 0 stays 0, everything else saturates = px * 65536; hugely negative alike.
__device__ __forceinline__ double row_factor_for_fma(double c) {
    if (c != c) return 0.0;
    if (c > 0x1p+960) return 65536.0;
    if (c < -0x1p+960) return -4294967296.0;
    return c;
}

struct HistSlicesArgs {
    shg::PtrBatch imgs;
    int64_t h, w, pitch;
    int tiles;
    int64_t th, tw;
    uint32_t* part;
    size_t zs;
    int slice_rows, vec;
    FusedSrc fs;
    int clip, chunk_rows;            // BITS < 16 only
};

// BITS = 16: a slice is fewer than 65536 pixels and its u16 counters are stored as they are (128 KB a slice).
// BITS = 8 / 4: SATURATED counters.  All CLAHE does with a tile's histogram is clip it -- min(count, clip), and the clipped-off excess
// area - sum of min(count, clip) -- and min(a + b, c) = min(min(a, c) + b, c) for counts, so a slice may forget everything above
// `clip` (12 at 2000 x 2098 px on a 2 x 2 grid, 20 at 2560 x 2675): the workgroup counts `chunk_rows` rows at a time (fewer than
// 65536 - 255 pixels: the u16 counters cannot wrap), clamps its counters to `clip` between chunks, and stores them as bytes
// (clip <= 255) or nibbles (clip <= 15): a slice of any length leaves 64 or 32 KB instead of 128 KB per 65535 pixels.
// The frame's highest order statistics survive the clamping too (hist_rank_top_job).
// a coordinate of the reflected border, i in [n, n + tiles): it mirrors once (clahe_impl requires n > tiles) -- without reflect101's
// 64-bit remainder, which a wave pays for in full when one lane of it is in the border
__device__ __forceinline__ int64_t mirror101(int64_t i, int64_t n) { return 2 * (n - 1) - i; }

template <bool FUSED, int BITS> __global__ __launch_bounds__(1024) void k_tile_hist16_slices(const HistSlicesArgs kargs) {
    const shg::PtrBatch& imgs = kargs.imgs;
    const int64_t h = kargs.h, w = kargs.w, pitch = kargs.pitch, th = kargs.th, tw = kargs.tw;
    const int tiles = kargs.tiles, slice_rows = kargs.slice_rows, vec = kargs.vec;
    uint32_t* __restrict__ part = kargs.part;
    const size_t zs = kargs.zs;
    const FusedSrc& fs = kargs.fs;
    extern __shared__ uint32_t lh[];   // HIST16/2 dwords, two u16 counters each; FUSED: then slice_rows doubles (the rows' factors)
    const uint16_t* __restrict__ img = imgs.at<const uint16_t>(blockIdx.z);
    part = zdisk(part, zs, blockIdx.z);
    const int tile = blockIdx.y;
    const int64_t ty = tile / tiles, tx = tile % tiles;
    // a slice is a run of whole tile rows (slice_rows of them: fewer than 65536 pixels, so that a u16 counter cannot wrap)
    const int ya = (int)blockIdx.x * slice_rows;
    const int yb = min((int)th, ya + slice_rows) - 1;
    for (int i = threadIdx.x; i < HIST16 / 8; i += 1024) reinterpret_cast<uint4*>(lh)[i] = make_uint4(0, 0, 0, 0);
    double* cf = reinterpret_cast<double*>(lh + HIST16 / 2);
    const uint16_t* __restrict__ raw = nullptr;
    uint16_t* __restrict__ fin = nullptr;
    __shared__ uint32_t fill_s;
    const bool scaled = FUSED && fs.c != nullptr;
    if (FUSED) {
        raw = fs.raw.at<const uint16_t>(blockIdx.z);
        fin = imgs.at<uint16_t>(blockIdx.z);
        // the factors of this slice's rows (they may live in the host's staging area: one trip for all of them, here) and the
        // value of the padding columns
        const double* c = scaled ? fs.c + (int64_t)blockIdx.z * h : nullptr;
        for (int i = threadIdx.x; i <= yb - ya; i += 1024) {
            int64_t r = ty * th + ya + i;
            if (r >= h) r = mirror101(r, h);           // a row of OpenCV's border: the factor of the row it mirrors
            cf[i] = scaled ? row_factor_for_fma(c[r]) : 1.0;
        }
        if (threadIdx.x == 0) fill_s = scaled ? scale_px(raw[0], c[0]) : (uint32_t)raw[0];
    }
    __syncthreads();
    const int chunk_rows = BITS == 16 ? slice_rows : kargs.chunk_rows;
    const uint32_t clip2 = (uint32_t)kargs.clip;
    for (int ca = ya; ca <= yb; ca += chunk_rows) {
        const int cb = min(yb, ca + chunk_rows - 1);
        if (BITS != 16 && ca != ya) {                        // between chunks: forget what lies above the clip limit
            __syncthreads();
            for (int i = threadIdx.x; i < HIST16 / 8; i += 1024) {
                uint4 q = reinterpret_cast<uint4*>(lh)[i];
                uint32_t* d = reinterpret_cast<uint32_t*>(&q);
#pragma unroll
                for (int j = 0; j < 4; ++j) d[j] = min(d[j] & 0xffffu, clip2) | (min(d[j] >> 16, clip2) << 16);
                reinterpret_cast<uint4*>(lh)[i] = q;
            }
            __syncthreads();
        }
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        const int twi = (int)tw;
        const int64_t xbase = tx * tw;
        auto count = [&](uint32_t v) { atomicAdd(&lh[v >> 1], 1u << ((v & 1u) << 4)); };
        if (vec) {
            // rows of 16-byte vectors: a wave takes two rows a round (its share of a 65535-pixel slice is four), three vectors of each
            // per lane, all six loads in flight before the first count -- pixel by pixel the wave waited for memory a dozen times per
            // slice, which is what this kernel's time was.  (FUSED with vec: no crop -- the frame's columns are the image's.)
            // The tile's columns of the image are [xa, xb): whole vectors from xa8 to xb8, and up to 7 + 7 pixels before and after them
            // (a tile 1049 pixels wide starts anywhere) plus the columns of the reflected border, [xb, xbase + tw) -- at most 30 pixels
            // a row, taken one per lane behind the chunk's vectors.
            const int64_t xa = xbase, xb = min(xbase + tw, w);
            const int64_t xa8 = (xa + 7) & ~(int64_t)7, xb8 = max(xb & ~(int64_t)7, xa8);
            const int nvr = (int)((xb8 - xa8) / 8);
            const int n_head = (int)(min(xa8, xb) - xa), n_tail = (int)(xb - max(xb8, xa + n_head)), n_edge = n_head + n_tail + (int)(xbase + tw - xb);
            // Round 6: the chunk's whole vectors as ONE flat sequence, (row, vector) = (i / nvr, i % nvr), four of them per thread in
            // flight.  Dealt row by row -- a wave two rows a round, three vector slots a lane -- a 1048-pixel tile (131 vectors) left the
            // third slot to 3 lanes of 64 and half of the second's second row idle: 12 slots a lane for 8 vectors, a third of the
            // count's instructions spent on masked-off lanes.
            {
                const uint32_t nvr_u = (uint32_t)nvr, total_v = (uint32_t)(cb - ca + 1) * nvr_u;
                const float inv_nvr = nvr > 0 ? 1.0f / (float)nvr : 0.0f;
                constexpr int U = 4;
                for (uint32_t i0 = threadIdx.x; i0 < total_v; i0 += 1024u * U) {
                    uint4 q[U];
                    int64_t yv[U];
                    uint32_t vv[U], rowv[U];
                    bool borderv[U];
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        const uint32_t i = i0 + 1024u * u < total_v ? i0 + 1024u * u : i0;
                        uint32_t r = (uint32_t)((float)i * inv_nvr);              // i / nvr, within one (i < 2^16): set right below
                        r -= (r * nvr_u > i) ? 1u : 0u;
                        r += ((r + 1u) * nvr_u <= i) ? 1u : 0u;
                        rowv[u] = r;
                        vv[u] = i - r * nvr_u;
                        int64_t y = ty * th + ca + (int64_t)r;
                        borderv[u] = y >= h;                                       // a row of the reflected border: counted, not stored
                        if (borderv[u]) y = mirror101(y, h);
                        yv[u] = y;
                        const uint16_t* srow = FUSED ? raw + y * fs.raw_pitch : img + y * pitch;
                        q[u] = reinterpret_cast<const uint4*>(srow + xa8)[vv[u]];
                    }
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        if (i0 + 1024u * u >= total_v) break;
                        uint32_t d[4] = {q[u].x, q[u].y, q[u].z, q[u].w};
                        if (FUSED) {
                            if (scaled) {
                                const double cy = cf[ca - ya + (int)rowv[u]], kc = cy * 0x1p+52;
#pragma unroll
                                for (int j = 0; j < 4; ++j) d[j] = (scale_px_fma(d[j] & 0xffffu, cy, kc) & 0xffffu) | (scale_px_fma(d[j] >> 16, cy, kc) << 16);
                            }
                            if (!borderv[u]) reinterpret_cast<uint4*>(fin + yv[u] * pitch + xa8)[vv[u]] = make_uint4(d[0], d[1], d[2], d[3]);
                        }
#pragma unroll
                        for (int j = 0; j < 4; ++j) { count(d[j] & 0xffffu); count(d[j] >> 16); }
                    }
                }
            }
            // the rows' loose ends, all of the chunk's together: one pixel a lane (behind every pair of rows they cost the whole wave a
            // pixel's instructions for a handful of lanes: 16 us of 116 over a 21-disk stack)
            for (int e = threadIdx.x; e < (cb - ca + 1) * n_edge; e += 1024) {
                const int rr = e / n_edge, l = e - rr * n_edge, yy = ca + rr;
                int64_t y = ty * th + yy;
                const bool border_row = y >= h;
                if (border_row) y = mirror101(y, h);
                const int64_t x = l < n_head ? xa + l : (l < n_head + n_tail ? xb - n_tail + (l - n_head) : xb + (l - n_head - n_tail));
                const bool border_col = x >= w;
                const int64_t xr = border_col ? mirror101(x, w) : x;
                uint32_t v = (FUSED ? raw + y * fs.raw_pitch : img + y * pitch)[xr];
                if (FUSED) {
                    const double cy = cf[yy - ya];
                    if (scaled) v = scale_px_fma(v, cy, cy * 0x1p+52) & 0xffffu;
                    if (!border_col && !border_row) fin[y * pitch + x] = (uint16_t)v;
                }
                count(v);
            }
        } else {
            for (int yy = ca + wave; yy <= cb; yy += 16) {
                // A grid that does not divide the image: OpenCV pads it below and to the right with its mirror image (REFLECT_101) and the
                // tiles count those pixels too.  They are pixels of the image -- formed here a second time where the image is being made
                // (FUSED), and counted, but not stored.
                int64_t y = ty * th + yy;
                const bool border_row = y >= h;
                if (border_row) y = mirror101(y, h);
                const uint16_t* row = FUSED ? raw + y * fs.raw_pitch : img + y * pitch;
                const double cy = FUSED ? cf[yy - ya] : 1.0, kc = cy * 0x1p+52;
                const uint32_t fill = FUSED ? fill_s : 0u;
                // eight loads in flight per lane before the first count: one load per trip left this kernel waiting on memory
                // latency 33 times over (20 us)
                for (int x0 = lane; x0 < twi; x0 += 64 * 8) {
                    uint32_t v[8];
                    bool inside[8], border[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int xx = x0 + 64 * u;
                        int64_t x = xbase + (xx < twi ? xx : 0);
                        border[u] = border_row || x >= w;
                        if (x >= w) x = mirror101(x, w);
                        if (FUSED) {
                            const int64_t j = x - fs.dx0;                      // the crop / pad block: column x of the image <- column sx0 + j of the frame
                            inside[u] = j >= 0 && j < fs.ncopy;
                            v[u] = row[inside[u] ? fs.sx0 + j : 0];
                        } else {
                            v[u] = row[x];
                        }
                    }
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        if (x0 + 64 * u >= twi) continue;
                        if (FUSED) {
                            v[u] = inside[u] ? (scaled ? scale_px_fma(v[u], cy, kc) & 0xffffu : v[u]) : fill;
                            if (!border[u]) fin[y * pitch + xbase + x0 + 64 * u] = (uint16_t)v[u];
                        }
                        count(v[u]);
                    }
                }
            }
        }
    }
    __syncthreads();
    if (BITS == 16) {
        uint32_t* out = part + ((int64_t)tile * gridDim.x + blockIdx.x) * (HIST16 / 2);
        for (int i = threadIdx.x; i < HIST16 / 8; i += 1024) reinterpret_cast<uint4*>(out)[i] = reinterpret_cast<const uint4*>(lh)[i];
    } else {
        // Stored word O folds the LDS words O, O + NW, ... (KF of them: NW = 32768 / KF stored words a slice), two fields each -- lanes
        // walk the LDS and the slice 16 bytes apart (a lane packing KF adjacent LDS quads read them 16 x KF bytes apart: 16 lanes on the
        // same banks, 7 us of a 19 us kernel).  So bin b sits in word (b / 2) % NW, field 2 * ((b / 2) / NW) + b % 2 (k_hist_reduce_sat).
        constexpr int KF = 16 / BITS;                        // 4 (nibbles), 2 (bytes)
        constexpr int NQ = HIST16 / 8 / KF;                  // stored quads a slice
        uint4* out = reinterpret_cast<uint4*>(part + ((int64_t)tile * gridDim.x + blockIdx.x) * (HIST16 * BITS / 32));
        for (int q = threadIdx.x; q < NQ; q += 1024) {
            uint32_t o[4] = {0, 0, 0, 0};
#pragma unroll
            for (int k = 0; k < KF; ++k) {
                const uint4 v = reinterpret_cast<const uint4*>(lh)[q + NQ * k];
                const uint32_t d[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] |= (min(d[j] & 0xffffu, clip2) | (min(d[j] >> 16, clip2) << BITS)) << (2 * BITS * k);
            }
            out[q] = make_uint4(o[0], o[1], o[2], o[3]);
        }
    }
}

// grid (32, ntiles) x 1024 threads: lane d of the tile owns the counter pair d = bins 2d, 2d + 1.
// hist [tile][65536] u32; chunk_tile [tile][1024] (64-bin sums); se [tile][32][2] = clipped total, excess per 2048 bins.
struct HistReduceArgs {
    const uint32_t* part;
    int slices, clip;
    uint32_t *hist, *chunk_tile;
    int32_t* se;
    size_t zs;
    uint32_t* sel_zero;
    int sel_words;
    int sel_window;                  // the window's counts and SelWin follow the sel_words words: cleared / rolled over as well
};
// the slot histograms of the selects that follow the blend (which adds to them) zeroed by the reduction's first workgroups instead of
// by a memset launch of their own (every launch is an L2 write-back and invalidate under the other scans' kernels); with a window:
// its counts too (workgroup x < SEL_SLOTS takes slot x's), and SelWin moves on: next scan's origin becomes this scan's, the maximum 0
__device__ __forceinline__ void zero_select_area(const HistReduceArgs& kargs, int nt) {
    if (!kargs.sel_zero || blockIdx.y != 0) return;
    uint32_t* z = zdisk(kargs.sel_zero, kargs.zs, blockIdx.z);
    if (blockIdx.x == 0)
        for (int i = threadIdx.x; i < kargs.sel_words; i += nt) z[i] = 0;
    if (kargs.sel_window) {
        uint32_t* win = z + kargs.sel_words;
        for (int slot = blockIdx.x; slot < SEL_WIN_SLOTS; slot += gridDim.x)
            for (int i = threadIdx.x; i < SEL_NW * 256; i += nt) win[slot * SEL_NW * 256 + i] = 0;
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            SelWin* st = reinterpret_cast<SelWin*>(win + SEL_WIN_WORDS);
            st->w0_cur = st->w0_next;
            st->max = 0;
        }
    }
}

__global__ __launch_bounds__(1024) void k_hist_reduce(const HistReduceArgs kargs) {
    const uint32_t* __restrict__ part = kargs.part;
    const int slices = kargs.slices, clip = kargs.clip;
    uint32_t* __restrict__ hist = kargs.hist;
    uint32_t* __restrict__ chunk_tile = kargs.chunk_tile;
    int32_t* __restrict__ se = kargs.se;
    const size_t zs = kargs.zs;
    __shared__ int wsum[2][16];
    zero_select_area(kargs, 1024);
    part = zdisk(part, zs, blockIdx.z);
    hist = zdisk(hist, zs, blockIdx.z);
    chunk_tile = zdisk(chunk_tile, zs, blockIdx.z);
    se = zdisk(se, zs, blockIdx.z);
    const int tile = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int d = blockIdx.x * 1024 + tid;
    const uint32_t* p = part + (int64_t)tile * slices * (HIST16 / 2) + d;
    uint32_t c0 = 0, c1 = 0;
    int s = 0;
    for (; s + 8 <= slices; s += 8) {
        uint32_t v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = p[(int64_t)(s + u) * (HIST16 / 2)];
#pragma unroll
        for (int u = 0; u < 8; ++u) { c0 += v[u] & 0xffffu; c1 += v[u] >> 16; }
    }
    for (; s < slices; ++s) {
        const uint32_t v = p[(int64_t)s * (HIST16 / 2)];
        c0 += v & 0xffffu;
        c1 += v >> 16;
    }
    *reinterpret_cast<uint2*>(hist + (int64_t)tile * HIST16 + 2 * d) = make_uint2(c0, c1);
    uint32_t t = c0 + c1;
    t = shg::sum_of_32_lanes(t);
    if ((lane & 31) == 0) chunk_tile[tile * 1024 + (d >> 5)] = t;
    int kept = (int)min(c0, (uint32_t)clip) + (int)min(c1, (uint32_t)clip);
    int over = (int)(c0 + c1) - kept;
    kept = shg::wave_sum(kept);
    over = shg::wave_sum(over);
    if (lane == 0) { wsum[0][wave] = kept; wsum[1][wave] = over; }
    __syncthreads();
    if (tid < 2) {
        int a = 0;
        for (int i = 0; i < 16; ++i) a += wsum[tid][i];
        se[(tile * 32 + blockIdx.x) * 2 + tid] = a;
    }
}

// The saturated slices of a tile added up, clamped again and stored as bytes: hist8 [tile][65536] = min(count, clip); kept512 [tile][128] =
// the clipped total of every 512 bins (the excess of the tile is its area minus their sum: k_tile_lut16_blocks<true>); chunk_tile
// [tile][1024] = the clamped counts of every 64 bins (hist_rank_top_job).
// grid (stored quads / 64, ntiles, disks) x 256 threads: lane l of every wave owns stored quad 64 * blockIdx.x + l -- KF runs of 8
// consecutive bins, see the packing in k_tile_hist16_slices -- and wave g adds the slices g, g + 4, ...; wave 0 adds the four up.
template <int BITS> __global__ __launch_bounds__(256) void k_hist_reduce_sat(const HistReduceArgs kargs) {
    constexpr int KF = 16 / BITS;
    constexpr int NQ = HIST16 / 8 / KF;
    constexpr int NA = BITS == 8 ? 8 : 16;                  // 16-bit sums, two to a register
    const int slices = kargs.slices;
    const uint32_t clip = (uint32_t)kargs.clip;
    const size_t zs = kargs.zs;
    __shared__ uint32_t partial[3][NA][64];
    zero_select_area(kargs, 256);
    const uint32_t* __restrict__ part = zdisk(kargs.part, zs, blockIdx.z);
    uint8_t* __restrict__ hist8 = reinterpret_cast<uint8_t*>(zdisk(kargs.hist, zs, blockIdx.z));
    uint32_t* __restrict__ chunk_tile = zdisk(kargs.chunk_tile, zs, blockIdx.z);
    int32_t* __restrict__ kept512 = zdisk(kargs.se, zs, blockIdx.z);
    const int tile = blockIdx.y, lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const int q = blockIdx.x * 64 + lane;
    const uint4* p = reinterpret_cast<const uint4*>(part + (int64_t)tile * slices * (NQ * 4)) + q;
    uint32_t acc[NA];
#pragma unroll
    for (int i = 0; i < NA; ++i) acc[i] = 0;
    auto add = [&](const uint4& v) {
        const uint32_t d[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (BITS == 8) {                                 // fields 0, 2 and 1, 3
                acc[2 * j] += d[j] & 0x00ff00ffu;
                acc[2 * j + 1] += (d[j] >> 8) & 0x00ff00ffu;
            } else {                                         // fields i and i + 4
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[4 * j + i] += (d[j] >> (4 * i)) & 0x000f000fu;
            }
        }
    };
    int s = grp;
    for (; s + 12 < slices; s += 16) {
        uint4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = p[(int64_t)(s + 4 * u) * NQ];
#pragma unroll
        for (int u = 0; u < 4; ++u) add(v[u]);
    }
    for (; s < slices; s += 4) add(p[(int64_t)s * NQ]);
    if (grp != 0) {
#pragma unroll
        for (int i = 0; i < NA; ++i) partial[grp - 1][i][lane] = acc[i];
    }
    __syncthreads();
    if (grp != 0) return;
#pragma unroll
    for (int i = 0; i < NA; ++i) acc[i] += partial[0][i][lane] + partial[1][i][lane] + partial[2][i][lane];
    // field f of stored word j (f = 2 k + half): bin 8 q + 16384 (8 / KF ... ) -- run k starts at bin 8 q + (HIST16 / KF) k, word j holds its bins 2 j, 2 j + 1
#pragma unroll
    for (int k = 0; k < KF; ++k) {
        uint32_t o[2] = {0, 0};
        int kept = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const int f = 2 * k + half;
                uint32_t c;
                if (BITS == 8) c = f < 2 ? (acc[2 * j + f] & 0xffffu) : (acc[2 * j + f - 2] >> 16);           // fields (0, 2) (1, 3)
                else c = f < 4 ? (acc[4 * j + f] & 0xffffu) : (acc[4 * j + f - 4] >> 16);                    // fields (i, i + 4)
                const uint32_t m = min(c, clip);
                kept += (int)m;
                const int bin8 = 2 * j + half;
                o[bin8 >> 2] |= m << (8 * (bin8 & 3));
            }
        }
        const int b0 = 8 * q + (HIST16 / KF) * k;
        *reinterpret_cast<uint2*>(hist8 + (int64_t)tile * HIST16 + b0) = make_uint2(o[0], o[1]);
        int t = kept;
        t = shg::sum_of_8_lanes(t);
        if ((lane & 7) == 0) chunk_tile[tile * 1024 + (b0 >> 6)] = (uint32_t)t;
#pragma unroll
        for (int d = 8; d <= 32; d <<= 1) t += __shfl_xor(t, d);
        if (lane == 0) kept512[tile * 128 + (b0 >> 9)] = t;
    }
}

// The tile LUT (clip, redistribute, prefix sum, scale: as k_tile_lut16_lds) by 32 workgroups per tile, 2048 bins each:
// the counts before a workgroup's first bin are the clipped totals of the workgroups before it, plus what the
// redistribution adds there -- `batch` per bin and one more for the bins 0, step, 2 step, ... below residual * step.
struct Ranks8 { int64_t v[8]; };            // the requested ranks travel as a kernel argument: no host-to-device copy per call

// The pixels OpenCV's reflected border adds to the tile histograms when the grid does not divide the image (tile_geometry): rows
// h .. he - 1 over the padded width, then columns w .. we - 1 of the image's own rows -- a few thousand mirror images of pixels near the
// bottom and the right edge.  The FRAME's order statistics, read off the same histograms, must leave them out again: the workgroup
// that finds a rank walks over them twice (their 64-bin runs, then the bins of the run picked) and subtracts.
struct BorderPx {
    shg::PtrBatch imgs;              // the finished images (the histogram kernel has run)
    int64_t h, w, pitch, he, we;     // he == h and we == w: no border
    __host__ __device__ bool any() const { return he != h || we != w; }
};
// (he - h <= tiles < h and we - w <= tiles < w -- clahe_impl requires it -- so a border coordinate mirrors once: 2 (n - 1) - i)
__device__ __forceinline__ uint32_t border_pixel(const BorderPx& bp, const uint16_t* __restrict__ img, int i, int below, int wide) {
    int y, x;
    if (i < below) { y = (int)bp.h + i / (int)bp.we; x = i % (int)bp.we; }
    else { const int j = i - below; y = j / wide; x = (int)bp.w + j % wide; }
    if (y >= (int)bp.h) y = 2 * ((int)bp.h - 1) - y;
    if (x >= (int)bp.w) x = 2 * ((int)bp.w - 1) - x;
    return img[(int64_t)y * bp.pitch + x];
}
// The walk over the border pixels, twice: a thread's first eight (all of them for a border of up to 8 x the workgroup's threads: 6 200
// of a 2000 x 2097 image on a 2 x 2 grid) are loaded together the first time -- eight dependent round trips otherwise, 10 us of a 15 us
// kernel -- and kept in registers for the second.
struct BorderRegs { uint32_t v[8]; };
template <bool FIRST, typename F>
__device__ __forceinline__ void for_each_border_pixel(const BorderPx& bp, const uint16_t* __restrict__ img, BorderRegs& regs, F f) {
    const int below = (int)((bp.he - bp.h) * bp.we), wide = max((int)(bp.we - bp.w), 1), n = below + (int)(bp.h * (bp.we - bp.w));
    const int nt = (int)blockDim.x, tid = (int)threadIdx.x;
    if (FIRST) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = tid + u * nt;
            regs.v[u] = i < n ? border_pixel(bp, img, i, below, wide) : ~0u;
        }
    }
#pragma unroll
    for (int u = 0; u < 8; ++u)
        if (regs.v[u] != ~0u) f(regs.v[u]);
    for (int i = tid + 8 * nt; i < n; i += nt) f(border_pixel(bp, img, i, below, wide));
}

// One order statistic of the image whose per-tile histograms CLAHE has built (a grid that does not divide the image: less the pixels
// of the reflected border, `bp`), by one workgroup of 1024: lane t takes the t-th run of 64 bins from the chunk sums, a
// workgroup scan finds the run that holds the rank, one wave scans its 64 bins.  (k_hist_ranks, and the extra workgroups of
// k_tile_lut16_blocks.)
__device__ __forceinline__ void hist_rank_job(const uint32_t* __restrict__ hist, const uint32_t* __restrict__ chunk_sums, int chunk_sets, int ntiles,
                                              int64_t rank, double* __restrict__ out, const BorderPx* bp = nullptr, const uint16_t* __restrict__ img = nullptr) {
    __shared__ int64_t wtot[16];
    __shared__ int64_t pick[2];
    __shared__ int border_n[1024];   // border pixels per 64-bin run, then (the first 64) per bin of the run picked
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool border = bp && bp->any();
    BorderRegs regs;
    int64_t local = 0;
    for (int k = 0; k < chunk_sets; ++k) local += chunk_sums[k * 1024 + tid];     // one set (k_chunk_sums) or one per tile (k_hist_reduce)
    if (border) {
        border_n[tid] = 0;
        __syncthreads();
        for_each_border_pixel<true>(*bp, img, regs, [&](uint32_t v) { atomicAdd(&border_n[v >> 6], 1); });
        __syncthreads();
        local -= border_n[tid];
    }
    int64_t incl = local;
    incl = shg::wave_scan(incl);
    if (lane == 63) wtot[wave] = incl;
    __syncthreads();
    for (int i = 0; i < wave; ++i) incl += wtot[i];
    if (incl - local <= rank && rank < incl) { pick[0] = tid; pick[1] = incl - local; }
    __syncthreads();
    const int64_t chunk = pick[0], below = pick[1];
    if (border) {
        if (tid < 64) border_n[tid] = 0;
        __syncthreads();
        for_each_border_pixel<false>(*bp, img, regs, [&](uint32_t v) { if ((int64_t)(v >> 6) == chunk) atomicAdd(&border_n[v & 63u], 1); });
        __syncthreads();
    }
    if (wave != 0) return;
    int64_t c = border ? -(int64_t)border_n[lane] : 0;
    for (int t = 0; t < ntiles; ++t) c += hist[(int64_t)t * HIST16 + chunk * 64 + lane];
    int64_t inc2 = c;
    inc2 = shg::wave_scan(inc2);
    if (below + inc2 - c <= rank && rank < below + inc2) *out = (double)(chunk * 64 + lane);
}

// The same order statistic off SATURATED histograms (hist8 [tile][65536] = min(count, clip) and their 64-bin sums, k_hist_reduce_sat):
// the K-th LARGEST pixel of the image, K <= clip.  Exact: walking down from the top bin, the true count above the answer's bin is
// below K <= clip, so no (tile, bin) up there was clamped and the clamped sums are the true ones; at the answer's bin the running sum
// reaches K whether that bin was clamped (a clamped bin alone holds clip >= K) or not.  One workgroup of 1024: lane t takes the 64-bin
// run 1023 - t, a workgroup scan finds the run that reaches K, wave 0 scans its bins from the top.
//
// With a reflected border (BorderPx) the clamped counts hold border pixels as well.  Taking them out again is exact as long as no (tile,
// bin) ABOVE the answer's bin was clamped -- which the walk checks on its way: the clamped sum above the bin it settles on, border pixels
// included, stays below `clip` (then every count up there is a true one; a clamped bin further up that made the walk pass the true
// answer by is above the bin it ends on, and seen).  Where it does not -- more than a handful of the frame's very brightest pixels mirrored
// in the border -- the answer is NaN and the caller selects over the image instead (shg_stage_process_frames).
__device__ __forceinline__ void hist_rank_top_job(const uint8_t* __restrict__ hist8, const uint32_t* __restrict__ chunk_tile, int ntiles, int64_t k_top,
                                                  double* __restrict__ out, int clip = 0, const BorderPx* bp = nullptr, const uint16_t* __restrict__ img = nullptr) {
    __shared__ int wtot[2][16];
    __shared__ int pick[3];
    __shared__ int border_n[1024];   // border pixels per 64-bin run (from the top), then (the first 64) per bin of the run picked
    __shared__ int first_s;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool border = bp && bp->any();
    BorderRegs regs;
    const int chunk = 1023 - tid;
    int with = 0;                                           // (`with` the border pixels: what the histograms hold)
    for (int t = 0; t < ntiles; t += 4) {                   // (four tiles' loads issued together: a 2 x 2 grid is one round trip)
        uint32_t v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = chunk_tile[min(t + u, ntiles - 1) * 1024 + chunk];
#pragma unroll
        for (int u = 0; u < 4; ++u) with += t + u < ntiles ? (int)v[u] : 0;
    }
    int local = with;
    if (border) {
        border_n[tid] = 0;
        if (tid == 0) { pick[0] = -1; first_s = 1024; }
        __syncthreads();
        for_each_border_pixel<true>(*bp, img, regs, [&](uint32_t v) { atomicAdd(&border_n[1023 - (int)(v >> 6)], 1); });
        __syncthreads();
        local -= border_n[tid];
    }
    int incl = local, incl_with = with;
    incl = shg::wave_scan(incl);
    incl_with = shg::wave_scan(incl_with);
    if (lane == 63) { wtot[0][wave] = incl; wtot[1][wave] = incl_with; }
    __syncthreads();
    for (int i = 0; i < wave; ++i) { incl += wtot[0][i]; incl_with += wtot[1][i]; }
    const int K = (int)k_top;
    const bool reached = incl - local < K && K <= incl;
    if (border) {
        // (sums that clamped bins and the border pixels taken out of them have made non-monotone can reach K in several runs: the
        // first from the top is the one the argument above is about)
        if (reached) atomicMin(&first_s, tid);
        __syncthreads();
        if (reached && tid == first_s) { pick[0] = chunk; pick[1] = incl - local; pick[2] = incl_with - with; }
    } else if (reached) { pick[0] = chunk; pick[1] = incl - local; pick[2] = incl_with - with; }
    __syncthreads();
    const int ch = pick[0], above = pick[1], above_with = pick[2];
    if (border) {
        if (ch < 0) {                                       // clamped counts that never reach K
            if (tid == 0) *out = __builtin_nan("");
            return;
        }
        if (tid < 64) border_n[tid] = 0;
        __syncthreads();
        for_each_border_pixel<false>(*bp, img, regs, [&](uint32_t v) { if ((int)(v >> 6) == ch) atomicAdd(&border_n[v & 63u], 1); });
        __syncthreads();
    }
    if (wave != 0) return;
    const int bin = ch * 64 + 63 - lane;
    int cw = 0;
    for (int t = 0; t < ntiles; t += 4) {
        uint32_t v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = hist8[(int64_t)min(t + u, ntiles - 1) * HIST16 + bin];
#pragma unroll
        for (int u = 0; u < 4; ++u) cw += t + u < ntiles ? (int)v[u] : 0;
    }
    const int c = border ? cw - border_n[63 - lane] : cw;
    int inc2 = c, inc2_with = cw;
    inc2 = shg::wave_scan(inc2);
    inc2_with = shg::wave_scan(inc2_with);
    const bool mine = above + inc2 - c < K && K <= above + inc2;
    if (!border) {
        if (mine) *out = (double)bin;
        return;
    }
    const unsigned long long hits = __ballot(mine);
    if (hits == 0) {
        if (lane == 0) *out = __builtin_nan("");
    } else if (lane == __ffsll((long long)hits) - 1) {
        *out = above_with + inc2_with - cw < clip ? (double)bin : __builtin_nan("");
    }
}

struct LutBlocksArgs {
    const uint32_t* hist;
    const int32_t* se;
    int clip;
    float lut_scale;
    uint16_t* lut;
    size_t zs;
    // the frame's order statistics read off the same histograms by n_ranks extra workgroups (blockIdx.x = 32 + rank index,
    // blockIdx.y = 0) instead of a launch of k_hist_ranks: they depend on nothing the LUT does
    int n_ranks;
    int64_t rank[2];
    const uint32_t* chunk_sums;
    int chunk_sets;
    double* ranks_out;
    int ranks_zstride;
    int tile_area;                   // SAT: the excess of a tile is its area minus its clipped total
    int ntiles;
    BorderPx border;                 // the rank workgroups: what the frame's order statistics leave out again
};

// SAT: hist is hist8 (k_hist_reduce_sat) and rank[] counts from the top (the rank[r]-th largest pixel, 1 = the maximum)
// ALLT: grid (32 + ranks, 1, disks) -- a workgroup builds its 2048 LUT entries for ALL the tiles, one after the other, and stores them
// together: entry (value, tile) sits at value * ntiles + tile, so a workgroup per tile wrote 2-byte pieces 2 * ntiles bytes apart (44 MB
// of write traffic for 11 MB of LUTs over a 21-disk stack); the 2048 x ntiles entries of a workgroup are one contiguous run.
// (ntiles <= 16: 64 KB of LDS; larger grids keep a workgroup per tile, grid (32 + ranks, ntiles, disks).)
template <bool SAT, bool ALLT> __global__ __launch_bounds__(1024) void k_tile_lut16_blocks(const LutBlocksArgs kargs) {
    const uint32_t* __restrict__ hist = kargs.hist;
    const int32_t* __restrict__ se = kargs.se;
    const int clip = kargs.clip;
    const float lut_scale = kargs.lut_scale;
    uint16_t* __restrict__ lut = kargs.lut;
    const size_t zs = kargs.zs;
    const int ntiles = kargs.ntiles;
    constexpr int HIST = 65536;
    __shared__ int s_before, s_excess;
    extern __shared__ uint16_t staged[];                     // ALLT: [2048][ntiles]
    hist = zdisk(hist, zs, blockIdx.z);
    if (blockIdx.x >= 32) {
        const int r = (int)blockIdx.x - 32;
        if (blockIdx.y == 0 && r < kargs.n_ranks) {
            double* out = kargs.ranks_out + (int64_t)blockIdx.z * kargs.ranks_zstride + r;
            const uint16_t* img = kargs.border.any() ? kargs.border.imgs.at<const uint16_t>(blockIdx.z) : nullptr;
            if (SAT) hist_rank_top_job(reinterpret_cast<const uint8_t*>(hist), zdisk(kargs.chunk_sums, zs, blockIdx.z), ntiles, kargs.rank[r], out, clip, &kargs.border, img);
            else hist_rank_job(hist, zdisk(kargs.chunk_sums, zs, blockIdx.z), kargs.chunk_sets, ntiles, kargs.rank[r], out, &kargs.border, img);
        }
        return;
    }
    se = zdisk(se, zs, blockIdx.z);
    lut = zdisk(lut, zs, blockIdx.z);
    __shared__ int wsum[16];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int first = b * 2048, i0 = first + 2 * tid;
    for (int tile = ALLT ? 0 : (int)blockIdx.y; tile < (ALLT ? ntiles : (int)blockIdx.y + 1); ++tile) {
        if (wave == 0) {
            int kept = 0, over = 0;
            if (SAT) {                                       // se = kept512 [tile][128]: the clipped totals of every 512 bins
                const int lo = se[tile * 128 + lane], hi = se[tile * 128 + 64 + lane];
                kept = (lane < 4 * b ? lo : 0) + (64 + lane < 4 * b ? hi : 0);
                over = lo + hi;
            } else if (lane < 32) {
                kept = lane < b ? se[(tile * 32 + lane) * 2] : 0;
                over = se[(tile * 32 + lane) * 2 + 1];
            }
            kept = shg::wave_sum(kept);
            over = shg::wave_sum(over);
            if (lane == 0) { s_before = kept; s_excess = SAT ? kargs.tile_area - over : over; }
        }
        uint2 hh;
        if (SAT) {
            const uint32_t two = *reinterpret_cast<const uint16_t*>(reinterpret_cast<const uint8_t*>(hist) + (int64_t)tile * HIST + i0);
            hh = make_uint2(two & 0xffu, two >> 8);
        } else {
            hh = *reinterpret_cast<const uint2*>(hist + (int64_t)tile * HIST + i0);
        }
        __syncthreads();
        const int excess = s_excess;
        const int batch = excess / HIST;
        const int residual = excess - batch * HIST;
        const int step = residual != 0 ? max(HIST / residual, 1) : 1;
        const int64_t limit = (int64_t)residual * step;
        int c0 = (int)min(hh.x, (uint32_t)clip) + batch, c1 = (int)min(hh.y, (uint32_t)clip) + batch;
        if (residual != 0) {
            if ((int64_t)i0 < limit && i0 % step == 0) c0 += 1;
            if ((int64_t)(i0 + 1) < limit && (i0 + 1) % step == 0) c1 += 1;
        }
        const int local = c0 + c1;
        int incl = local;
        incl = shg::wave_scan(incl);
        if (lane == 63) wsum[wave] = incl;
        __syncthreads();
        int base = 0;
        for (int i = 0; i < wave; ++i) base += wsum[i];
        const int64_t reach = (int64_t)first < limit ? (int64_t)first : limit;
        const int bumped = residual != 0 ? (int)((reach + step - 1) / step) : 0;      // bins 0, step, ... before `first`
        const int run0 = s_before + batch * first + bumped + base + incl - local + c0;
        const int run1 = run0 + c1;
        int r0 = __float2int_rn(__int2float_rn(run0) * lut_scale);                    // saturate_cast<T>(sum * lutScale)
        int r1 = __float2int_rn(__int2float_rn(run1) * lut_scale);
        r0 = r0 < 0 ? 0 : (r0 > HIST - 1 ? HIST - 1 : r0);
        r1 = r1 < 0 ? 0 : (r1 > HIST - 1 ? HIST - 1 : r1);
        // stored value-major, [value][tile]: the (up to) four tile LUT entries a pixel blends sit side by side (k_clahe_interp_vm)
        if (ALLT) {
            staged[2 * tid * ntiles + tile] = (uint16_t)r0;
            staged[(2 * tid + 1) * ntiles + tile] = (uint16_t)r1;
            __syncthreads();                                 // (also: s_before / s_excess / wsum are the next tile's now)
        } else {
            lut[(int64_t)i0 * ntiles + tile] = (uint16_t)r0;
            lut[(int64_t)(i0 + 1) * ntiles + tile] = (uint16_t)r1;
        }
    }
    if (ALLT) {
        const int words = 2048 * ntiles / 2;                 // the run as 32-bit words (2048 entries x ntiles: even)
        uint32_t* dst = reinterpret_cast<uint32_t*>(lut + (int64_t)first * ntiles);
        const uint32_t* src = reinterpret_cast<const uint32_t*>(staged);
        for (int i = tid; i < words; i += 1024) dst[i] = src[i];
    }
}

// 8-bit images: 256 bins, LDS-private u32 histogram
__global__ __launch_bounds__(256) void k_tile_hist8(const uint8_t* __restrict__ img, int64_t h, int64_t w, int64_t pitch,
                                                    int tiles, int64_t th, int64_t tw, uint32_t* __restrict__ hist) {
    __shared__ uint32_t lh[256];
    const int tile = blockIdx.y;
    const int64_t ty = tile / tiles, tx = tile % tiles;
    const int64_t area = th * tw;
    lh[threadIdx.x] = 0;
    __syncthreads();
    for (int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x; p < area; p += (int64_t)gridDim.x * 256) {
        const int64_t yy = p / tw, xx = p - yy * tw;
        atomicAdd(&lh[ext_pixel(img, h, w, pitch, ty * th + yy, tx * tw + xx)], 1u);
    }
    __syncthreads();
    if (lh[threadIdx.x]) atomicAdd(&hist[(int64_t)tile * 256 + threadIdx.x], lh[threadIdx.x]);
}

// One workgroup per tile: clip, redistribute, prefix-sum, scale.  1024 lanes x (hist/1024) bins.
template <int HIST>
__global__ __launch_bounds__(1024) void k_tile_lut(const uint32_t* __restrict__ hist, int clip, float lut_scale,
                                                   uint16_t* __restrict__ lut) {
    constexpr int PER = HIST >= 1024 ? HIST / 1024 : 1;
    constexpr int ACTIVE = HIST >= 1024 ? 1024 : HIST;
    __shared__ int wsum[16];
    __shared__ int total_clipped;
    const int tile = blockIdx.x;
    const uint32_t* hin = hist + (int64_t)tile * HIST;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const bool active = tid < ACTIVE;
    int bins[PER];
    int clipped = 0;
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        int c = active ? (int)hin[tid * PER + j] : 0;
        if (clip > 0 && c > clip) { clipped += c - clip; c = clip; }
        bins[j] = c;
    }
    // workgroup total of the clipped excess
    int v = clipped;
    v = shg::wave_sum(v);
    if (lane == 0) wsum[wave] = v;
    __syncthreads();
    if (tid == 0) {
        int s = 0;
        for (int i = 0; i < 16; ++i) s += wsum[i];
        total_clipped = s;
    }
    __syncthreads();
    const int excess = total_clipped;
    int local = 0;
    if (clip > 0) {
        const int batch = excess / HIST;
        const int residual = excess - batch * HIST;
        const int step = residual != 0 ? max(HIST / residual, 1) : 1;
        // for (i = 0; i < histSize && residual > 0; i += step, residual--) hist[i]++ : bins 0, step, 2 step, ... below
        // residual * step; one division per lane instead of two per bin
        const int64_t limit = (int64_t)residual * step;
        int next = residual != 0 ? ((tid * PER + step - 1) / step) * step : HIST;
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            const int i = tid * PER + j;
            int c = bins[j] + batch;
            if (i == next) {
                if ((int64_t)i < limit) c += 1;
                next += step;
            }
            bins[j] = c;
        }
    }
#pragma unroll
    for (int j = 0; j < PER; ++j) local += bins[j];
    // exclusive scan of `local` across the workgroup
    int incl = local;
    incl = shg::wave_scan(incl);
    __syncthreads();
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    int base = 0;
    for (int i = 0; i < wave; ++i) base += wsum[i];
    int run = base + incl - local;
    if (active) {
        uint16_t* lout = lut + (int64_t)tile * HIST;
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            run += bins[j];
            int r = __float2int_rn(__int2float_rn(run) * lut_scale);     // saturate_cast<T>(sum * lutScale)
            r = r < 0 ? 0 : (r > HIST - 1 ? HIST - 1 : r);
            lout[tid * PER + j] = (uint16_t)r;
        }
    }
}

// The same tile LUT for 65536 bins with coalesced global traffic.  In k_tile_lut a lane owns 64 consecutive bins, so
// every wave-load touches 64 cache lines (27 us per image).  Here the histogram is read lane-contiguously, clipped and
// parked in LDS as uint16 (needs 0 < clip <= 65535; index padded by one per 64 so that the blocked re-read -- lane t
// takes bins 64t .. 64t+63 -- spreads over the banks), processed exactly as above, and the LUT goes back through the
// same LDS array to be stored lane-contiguously.
__global__ __launch_bounds__(1024) void k_tile_lut16_lds(const uint32_t* __restrict__ hist, int clip, float lut_scale,
                                                         uint16_t* __restrict__ lut) {
    constexpr int HIST = 65536, PER = 64;
    extern __shared__ uint16_t cb[];                      // [HIST + HIST / 64]
    __shared__ int wsum[16];
    __shared__ int total_clipped;
    const int tile = blockIdx.x;
    const uint32_t* hin = hist + (int64_t)tile * HIST;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    int clipped = 0;
    for (int j0 = 0; j0 < PER; j0 += 16) {              // sixteen loads in flight per lane
        int c[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) c[u] = (int)hin[(j0 + u) * 1024 + tid];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int i = (j0 + u) * 1024 + tid;
            if (c[u] > clip) { clipped += c[u] - clip; c[u] = clip; }
            cb[i + (i >> 6)] = (uint16_t)c[u];
        }
    }
    int v = clipped;
    v = shg::wave_sum(v);
    if (lane == 0) wsum[wave] = v;
    __syncthreads();
    if (tid == 0) {
        int s = 0;
        for (int i = 0; i < 16; ++i) s += wsum[i];
        total_clipped = s;
    }
    __syncthreads();
    const int excess = total_clipped;
    const int batch = excess / HIST;
    const int residual = excess - batch * HIST;
    const int step = residual != 0 ? max(HIST / residual, 1) : 1;
    int bins[PER];
    int local = 0;
    // for (i = 0; i < histSize && residual > 0; i += step, residual--) hist[i]++ : the bins i = 0, step, 2 step, ...
    // below residual * step.  One division per lane finds the first such bin of its range (a division per bin made this
    // kernel ALU bound: 2 x 65536 integer divisions on one CU).
    const int64_t limit = (int64_t)residual * step;
    int next = residual != 0 ? ((tid * PER + step - 1) / step) * step : HIST;
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        const int i = tid * PER + j;
        int c = (int)cb[tid * (PER + 1) + j] + batch;
        if (i == next) {
            if ((int64_t)i < limit) c += 1;
            next += step;
        }
        bins[j] = c;
        local += c;
    }
    int incl = local;
    incl = shg::wave_scan(incl);
    __syncthreads();
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    int base = 0;
    for (int i = 0; i < wave; ++i) base += wsum[i];
    int run = base + incl - local;
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        run += bins[j];
        int r = __float2int_rn(__int2float_rn(run) * lut_scale);         // saturate_cast<T>(sum * lutScale)
        r = r < 0 ? 0 : (r > HIST - 1 ? HIST - 1 : r);
        cb[tid * (PER + 1) + j] = (uint16_t)r;
    }
    __syncthreads();
    uint16_t* lout = lut + (int64_t)tile * HIST;
#pragma unroll 8
    for (int j = 0; j < PER; ++j) {
        const int i = j * 1024 + tid;
        lout[i] = cb[i + (i >> 6)];
    }
}

template <typename T, int HIST>
__global__ __launch_bounds__(256) void k_clahe_interp(const T* __restrict__ img, int64_t h, int64_t w, int64_t pitch,
                                                      int tiles, float inv_tw, float inv_th,
                                                      const uint16_t* __restrict__ lut, T* __restrict__ dst, int64_t dst_pitch) {
    const int64_t x = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t y = blockIdx.y;
    if (x >= w) return;
    const float txf = (float)(int)x * inv_tw - 0.5f;
    int tx1 = (int)floorf(txf);
    int tx2 = tx1 + 1;
    const float xa = txf - (float)tx1;
    const float xa1 = 1.0f - xa;
    tx1 = max(tx1, 0);
    tx2 = min(tx2, tiles - 1);
    const float tyf = (float)(int)y * inv_th - 0.5f;
    int ty1 = (int)floorf(tyf);
    int ty2 = ty1 + 1;
    const float ya = tyf - (float)ty1;
    const float ya1 = 1.0f - ya;
    ty1 = max(ty1, 0);
    ty2 = min(ty2, tiles - 1);
    const int v = img[y * pitch + x];
    const uint16_t* p1 = lut + (int64_t)(ty1 * tiles) * HIST + v;
    const uint16_t* p2 = lut + (int64_t)(ty2 * tiles) * HIST + v;
    const float l11 = (float)(int)p1[(int64_t)tx1 * HIST], l12 = (float)(int)p1[(int64_t)tx2 * HIST];
    const float l21 = (float)(int)p2[(int64_t)tx1 * HIST], l22 = (float)(int)p2[(int64_t)tx2 * HIST];
    const float res = (l11 * xa1 + l12 * xa) * ya1 + (l21 * xa1 + l22 * xa) * ya;
    int r = __float2int_rn(res);
    r = r < 0 ? 0 : (r > HIST - 1 ? HIST - 1 : r);
    dst[y * dst_pitch + x] = (T)r;
}

// The same blend from a value-major LUT, lut[value][tile] (k_tile_lut16_blocks).  k_clahe_interp is bound by the rate
// of its LUT reads: four 2-byte gathers per pixel, each lane of each one in a cache line of its own (neighbouring
// pixels differ by more than the 32 values a line holds), 16.8 M line requests per image, 18 us.  Value-major, a
// pixel's four entries share a line; with the reference's 2 x 2 grid they are one aligned 8-byte word.
// PX pixels per lane (4: rows 8-byte aligned, one 8-byte load and store); a workgroup takes `rows` rounds of 256 lanes.
// COUNT: also the first pass of the order statistics that follow (np.percentile(cl1, 10), np.max(cl1)) -- the histogram
// of the high bytes of the pixels it has just produced, in sel_hist's slot layout (k_select16_pass, pass 0): the values
// are in registers here, which saves that pass its read of the image.
struct InterpVmArgs {
    shg::PtrBatch imgs;
    int64_t h, w, pitch;
    int tiles;
    float inv_tw, inv_th;
    const uint16_t* lut;
    shg::PtrBatch dsts;
    int64_t dst_pitch;
    int rows;
    uint32_t* sel_hist;
    int sel_stride;
    size_t zs;
    int tiled;
    uint32_t tiles_x;
    int sel_window;                  // COUNT: also the low bytes inside the window and the largest pixel (SelWin behind sel_hist's slots)
};

// T2: a 2 x 2 tile grid, known at compile time.  With `tiles` a run-time value the four-entry load sat inside a branch per pixel and
// the compiler waited for each before issuing the next: a lane's PX gathers went one after the other, eight L2 round trips a row
// (the kernel's 0.18 of the roofline; its instruction count never mattered).  With T2 all PX gathers are issued, then blended.
constexpr int INTERP_ROUNDS = 4;                              // rounds of a workgroup in the tiled walk (launch_interp16)
template <int PX, bool COUNT, bool T2> __global__ __launch_bounds__(256) void k_clahe_interp_vm(const InterpVmArgs kargs) {
    const shg::PtrBatch& imgs = kargs.imgs;
    const shg::PtrBatch& dsts = kargs.dsts;
    const int64_t h = kargs.h, w = kargs.w, pitch = kargs.pitch, dst_pitch = kargs.dst_pitch;
    const int tiles = kargs.tiles, rows = kargs.rows, sel_stride = kargs.sel_stride, tiled = kargs.tiled;
    const float inv_tw = kargs.inv_tw, inv_th = kargs.inv_th;
    const uint16_t* __restrict__ lut = kargs.lut;
    uint32_t* __restrict__ sel_hist = kargs.sel_hist;
    const size_t zs = kargs.zs;
    const uint32_t tiles_x = kargs.tiles_x;
    constexpr int HIST = 65536;
    constexpr int COPIES = 8;                            // interleaved copies of each bin: a row's pixels crowd a few bins
    __shared__ uint32_t lh[COUNT ? 256 * COPIES : 1];
    __shared__ uint32_t lwin[COUNT ? SEL_NW * 256 : 1];
    __shared__ uint32_t wmax_s[4];
    const uint16_t* __restrict__ img = imgs.at<const uint16_t>(blockIdx.z);
    uint16_t* __restrict__ dst = dsts.at<uint16_t>(blockIdx.z);
    lut = zdisk(lut, zs, blockIdx.z);
    if (COUNT) sel_hist = zdisk(sel_hist, zs, blockIdx.z);
    const bool window = COUNT && kargs.sel_window;
    uint32_t* win = nullptr;
    SelWin* wstate = nullptr;
    uint32_t w0 = 0, vmax = 0;
    if (COUNT) {
        for (int i = threadIdx.x; i < 256 * COPIES; i += 256) lh[i] = 0;
        if (window) {
            win = sel_hist + (int64_t)SEL_SLOTS * sel_stride;
            wstate = reinterpret_cast<SelWin*>(win + SEL_WIN_WORDS);
            w0 = sel_window_origin(wstate);
            for (int i = threadIdx.x; i < SEL_NW * 256; i += 256) lwin[i] = 0;
        }
        __syncthreads();
    }
    // lanes are dealt (row, vector) pairs in one flat sequence, `rows` rounds of 256 per workgroup: with an (x, y) grid a width
    // just past a multiple of 256 * PX pixels (2096 at C2) leaves every third workgroup with a dozen lanes to do
    const uint32_t nv = (uint32_t)((w + PX - 1) / PX);
    const int ntiles = tiles * tiles;
    const int copy = threadIdx.x & (COPIES - 1);
    // What depends on the column alone -- the two tile columns a pixel blends and their weights -- is formed once per lane where the
    // lane keeps its columns over the `rows` rounds (the tiled walk: a third of this kernel's instructions were spent redoing it
    // for every row; the kernel is bound by its VALU work, 57 instructions a pixel before).
    float xa[PX], xa1[PX];
    int tx1[PX], tx2[PX];
    auto column_terms = [&](int64_t x0) {
#pragma unroll
        for (int j = 0; j < PX; ++j) {
            const float txf = (float)(int)(x0 + j) * inv_tw - 0.5f;
            const int t1 = (int)floorf(txf);
            xa[j] = txf - (float)t1;
            xa1[j] = 1.0f - xa[j];
            tx1[j] = max(t1, 0);
            tx2[j] = min(t1 + 1, tiles - 1);
            if (T2) { tx1[j] *= 16; tx2[j] *= 16; }          // (2 x 2 grid: the entry's bit offset inside its tile row's word)
        }
    };
    int64_t x0 = 0;
    uint32_t row0 = 0, row_step = 0;
    if (tiled) {
        // a wave takes 16 pixels x 16 rows, the four waves of a workgroup sit side by side (128 bytes of every row): the
        // pixels of a wave -- and of the wave that follows it on the CU -- are neighbours in both directions, so the window
        // of the LUT they read is a fifth of what 256 pixels along one row span, and more of its lines are still in L1
        const uint32_t bx = blockIdx.x % tiles_x, by = blockIdx.x / tiles_x;
        const uint32_t lw = (uint32_t)tiled & 0xffu, wx = ((uint32_t)tiled >> 8) & 0xffu;     // log2 of lanes across a wave, waves across
        const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
        const uint32_t lx = lane & ((1u << lw) - 1u), ly = lane >> lw;
        const uint32_t wvx = wave & ((1u << wx) - 1u), wvy = wave >> wx;
        const uint32_t wave_rows = 64u >> lw, wg_rows = wave_rows * (4u >> wx);
        x0 = ((int64_t)bx * ((1u << lw) << wx) + (wvx << lw) + lx) * PX;
        row0 = by * (uint32_t)rows * wg_rows + wvy * wave_rows + ly;
        row_step = wg_rows;
        if (x0 < w) column_terms(x0);
    }
    // (tiled walk, whole vectors: the NEXT round's pixels are asked for as soon as this round's have been unpacked, so that a round is
    // one round trip -- its gathers -- and not two)
    const bool ahead = tiled && (PX == 8 || PX == 4) && x0 + PX <= w && rows <= INTERP_ROUNDS;
    auto fetch = [&](uint32_t yy) {
        if (PX == 8) return *reinterpret_cast<const uint4*>(img + (int64_t)yy * pitch + x0);
        const uint2 t = *reinterpret_cast<const uint2*>(img + (int64_t)yy * pitch + x0);
        return make_uint4(t.x, t.y, 0u, 0u);
    };
    // Round 6: ALL of the workgroup's rounds asked for up front (the launch hands a tiled workgroup at most INTERP_ROUNDS of them): with one
    // round ahead a lane had one 16-byte load in flight -- 20 KB a CU, 2.6 TB/s at best: without its gathers, stores and arithmetic the
    // kernel still took 63 us for 67 Mpx (tools/interp_harness.py, profiles/r06_sweeps.txt).
    uint4 qs[INTERP_ROUNDS];
#pragma unroll
    for (int k = 0; k < INTERP_ROUNDS; ++k) {
        const uint32_t yk = row0 + (uint32_t)k * row_step;
        qs[k] = make_uint4(0u, 0u, 0u, 0u);
        if (ahead && k < rows && (int64_t)yk < h) qs[k] = fetch(yk);
    }
    // the rest of a round once the pixels' LUT entries are here: blend, store, the select's counts
    auto finish = [&](int64_t y, int n, int ty1, int ty2, float ya, float ya1, const uint32_t (&px)[PX], const uint2 (&q4)[PX]) {
        uint32_t out[PX];
#pragma unroll
        for (int j = 0; j < PX; ++j) {
            uint32_t l11, l12, l21, l22;
            if (T2) {
                const uint32_t top = ty1 ? q4[j].y : q4[j].x, bot = ty2 ? q4[j].y : q4[j].x;
                l11 = (top >> tx1[j]) & 0xffffu;
                l12 = (top >> tx2[j]) & 0xffffu;
                l21 = (bot >> tx1[j]) & 0xffffu;
                l22 = (bot >> tx2[j]) & 0xffffu;
            } else {
                const uint16_t* e = lut + (int64_t)px[j] * ntiles;
                l11 = e[ty1 * tiles + tx1[j]];
                l12 = e[ty1 * tiles + tx2[j]];
                l21 = e[ty2 * tiles + tx1[j]];
                l22 = e[ty2 * tiles + tx2[j]];
            }
            const float res = ((float)(int)l11 * xa1[j] + (float)(int)l12 * xa[j]) * ya1 + ((float)(int)l21 * xa1[j] + (float)(int)l22 * xa[j]) * ya;
            const int r = __float2int_rn(res);
            out[j] = (uint32_t)(r < 0 ? 0 : (r > HIST - 1 ? HIST - 1 : r));
        }
        if (PX == 8 && n == 8) {
            *reinterpret_cast<uint4*>(dst + y * dst_pitch + x0) = make_uint4(out[0] | (out[1 % PX] << 16), out[2 % PX] | (out[3 % PX] << 16),
                                                                             out[4 % PX] | (out[5 % PX] << 16), out[6 % PX] | (out[7 % PX] << 16));
        } else if (PX == 4 && n == 4) {
            *reinterpret_cast<uint2*>(dst + y * dst_pitch + x0) = make_uint2(out[0] | (out[1 % PX] << 16), out[2 % PX] | (out[3 % PX] << 16));
        } else {
#pragma unroll
            for (int j = 0; j < PX; ++j)
                if (j < n) dst[y * dst_pitch + x0 + j] = (uint16_t)out[j];
        }
        if (COUNT) {
            const uint32_t b0 = out[0] >> 8;
            bool same = n == PX;
#pragma unroll
            for (int j = 1; j < PX; ++j) same = same && (out[j] >> 8) == b0;
            if (same) {                                  // neighbours mostly share their high byte: one atomic for the lane
                atomicAdd(&lh[b0 * COPIES + copy], (uint32_t)PX);
            } else {
#pragma unroll
                for (int j = 0; j < PX; ++j)
                    if (j < n) atomicAdd(&lh[(out[j] >> 8) * COPIES + copy], 1u);
            }
            if (window) {
                // (the sky is one flat value over most of a wave -- and it is where the 10th percentile lies: 64 x PX additions to one
                // LDS word would queue up behind each other, so a wave whose pixels are all one value adds them up in one lane)
                const uint32_t first = __builtin_amdgcn_readfirstlane(out[0]);
                bool one_value = n == PX;
#pragma unroll
                for (int j = 1; j < PX; ++j) one_value = one_value && out[j] == out[0];
                const unsigned long long act = __ballot(1), eq = __ballot(one_value && out[0] == first);
                if (eq == act) {
                    const uint32_t d = (first >> 8) - w0;
                    if (d < (uint32_t)SEL_NW && (threadIdx.x & 63) == (unsigned)(__ffsll((long long)act) - 1))
                        atomicAdd(&lwin[d * 256 + (first & 0xffu)], (uint32_t)PX * (uint32_t)__popcll(act));
                    vmax = first > vmax ? first : vmax;
                } else {
#pragma unroll
                    for (int j = 0; j < PX; ++j) {
                        if (j < n) {
                            const uint32_t d = (out[j] >> 8) - w0;
                            if (d < (uint32_t)SEL_NW) atomicAdd(&lwin[d * 256 + (out[j] & 0xffu)], 1u);
                            vmax = out[j] > vmax ? out[j] : vmax;
                        }
                    }
                }
            }
        }
    };
    auto row_terms = [&](int64_t y, int& ty1, int& ty2, float& ya, float& ya1) {
        const float tyf = (float)(int)y * inv_th - 0.5f;
        ty1 = (int)floorf(tyf);
        ty2 = ty1 + 1;
        ya = tyf - (float)ty1;
        ya1 = 1.0f - ya;
        ty1 = max(ty1, 0);
        ty2 = min(ty2, tiles - 1);
    };
    if (T2 && ahead) {
        // Round 6: the gathers of the NEXT round are in flight while this one is blended (two sets of entry registers): a round's chain was
        // pixels -> 8 gathers -> wait -> blend, four times over; one round trip to L2 per round was the kernel (tools/interp_harness.py:
        // without its gathers 87 us against 152 for 67 Mpx, the blend's arithmetic 2.5 of them).
        int rounds_here = 0;
        if ((int64_t)row0 < h) rounds_here = (int)min((int64_t)rows, (h - (int64_t)row0 + row_step - 1) / row_step);
        auto issue = [&](int k, uint2 (&q)[PX]) {
            uint4 v = qs[0];                                   // (picked by compares: an indexed register array would go to scratch)
#pragma unroll
            for (int kk = 1; kk < INTERP_ROUNDS; ++kk)
                if (k == kk) v = qs[kk];
            uint32_t p[PX];
            p[0] = v.x & 0xffffu; p[1 % PX] = v.x >> 16; p[2 % PX] = v.y & 0xffffu; p[3 % PX] = v.y >> 16;
            if (PX == 8) { p[4 % PX] = v.z & 0xffffu; p[5 % PX] = v.z >> 16; p[6 % PX] = v.w & 0xffffu; p[7 % PX] = v.w >> 16; }
#pragma unroll
            for (int j = 0; j < PX; ++j) q[j] = *reinterpret_cast<const uint2*>(lut + (int64_t)p[j] * 4);
        };
        auto round_of = [&](int k, const uint2 (&q)[PX]) {
            const int64_t y = row0 + (uint32_t)k * row_step;
            int ty1, ty2;
            float ya, ya1;
            row_terms(y, ty1, ty2, ya, ya1);
            const uint32_t none[PX] = {};
            finish(y, PX, ty1, ty2, ya, ya1, none, q);
        };
        uint2 qa[PX], qb[PX];
        if (rounds_here > 0) issue(0, qa);
        for (int it = 0; it < rounds_here; it += 2) {
            if (it + 1 < rounds_here) issue(it + 1, qb);
            round_of(it, qa);
            if (it + 1 < rounds_here) {
                if (it + 2 < rounds_here) issue(it + 2, qa);
                round_of(it + 1, qb);
            }
        }
    } else
    for (int it = 0; it < rows; ++it) {
        uint32_t yy;
        if (tiled) {
            yy = row0 + (uint32_t)it * row_step;
            if (x0 >= w) break;
        } else {
            const uint32_t flat = (blockIdx.x * (uint32_t)rows + (uint32_t)it) * 256u + threadIdx.x;
            yy = flat / nv;
            x0 = (int64_t)(flat - yy * nv) * PX;
            column_terms(x0);
        }
        const int64_t y = yy;
        if (y >= h) break;
        const int n = (int)min((int64_t)PX, w - x0);
        int ty1, ty2;
        float ya, ya1;
        row_terms(y, ty1, ty2, ya, ya1);
        uint32_t px[PX];
        if (ahead) {
            uint4 q = qs[0];                                   // (picked by compares: an indexed register array would go to scratch)
#pragma unroll
            for (int k = 1; k < INTERP_ROUNDS; ++k)
                if (it == k) q = qs[k];
            px[0] = q.x & 0xffffu; px[1 % PX] = q.x >> 16; px[2 % PX] = q.y & 0xffffu; px[3 % PX] = q.y >> 16;
            if (PX == 8) { px[4 % PX] = q.z & 0xffffu; px[5 % PX] = q.z >> 16; px[6 % PX] = q.w & 0xffffu; px[7 % PX] = q.w >> 16; }
        } else if (PX == 8 && n == 8) {
            const uint4 q = *reinterpret_cast<const uint4*>(img + y * pitch + x0);
            px[0] = q.x & 0xffffu; px[1 % PX] = q.x >> 16; px[2 % PX] = q.y & 0xffffu; px[3 % PX] = q.y >> 16;
            px[4 % PX] = q.z & 0xffffu; px[5 % PX] = q.z >> 16; px[6 % PX] = q.w & 0xffffu; px[7 % PX] = q.w >> 16;
        } else if (PX == 4 && n == 4) {
            const uint2 q = *reinterpret_cast<const uint2*>(img + y * pitch + x0);
            px[0] = q.x & 0xffffu; px[1 % PX] = q.x >> 16; px[2 % PX] = q.y & 0xffffu; px[3 % PX] = q.y >> 16;
        } else {
#pragma unroll
            for (int j = 0; j < PX; ++j) px[j] = j < n ? img[y * pitch + x0 + j] : 0;
        }
        uint2 q4[PX] = {};                                   // T2: the pixel's four entries are one 8-byte load: [tile row 0: columns 0, 1 | tile row 1: columns 0, 1]
        if (T2) {
#pragma unroll
            for (int j = 0; j < PX; ++j) q4[j] = *reinterpret_cast<const uint2*>(lut + (int64_t)px[j] * 4);
        }
        finish(y, n, ty1, ty2, ya, ya1, px, q4);
    }
    if (COUNT) {
        if (window) {
            vmax = shg::wave_fold_u32(vmax, [](uint32_t x, uint32_t y) { return x > y ? x : y; });
            if ((threadIdx.x & 63) == 0) wmax_s[threadIdx.x >> 6] = vmax;
        }
        __syncthreads();
        uint32_t c = 0;
#pragma unroll
        for (int k = 0; k < COPIES; ++k) c += lh[threadIdx.x * COPIES + k];
        const unsigned slot = blockIdx.x % SEL_SLOTS;
        if (c) atomicAdd(&sel_hist[(int64_t)slot * sel_stride + threadIdx.x], c);
        if (window) {
            for (int i = threadIdx.x; i < SEL_NW * 256; i += 256) {
                const uint32_t cw = lwin[i];
                if (cw) atomicAdd(&win[(int64_t)(blockIdx.x % SEL_WIN_SLOTS) * SEL_NW * 256 + i], cw);
            }
            if (threadIdx.x == 0) {
                const uint32_t m = max(max(wmax_s[0], wmax_s[1]), max(wmax_s[2], wmax_s[3]));
                if (m) atomicMax(&wstate->max, m);
            }
        }
    }
}

// whole-image histogram (65536 or 256 bins) with the same LDS privatisation
__global__ __launch_bounds__(1024) void k_image_hist16(const uint16_t* __restrict__ img, int64_t h, int64_t w, int64_t pitch,
                                                       uint32_t* __restrict__ hist) {
    extern __shared__ uint32_t lh[];
    const int64_t n = h * w;
    const int64_t p0 = (int64_t)blockIdx.x * SLICE_PX;
    const int64_t p1 = p0 + SLICE_PX < n ? p0 + SLICE_PX : n;
    for (int i = threadIdx.x; i < HIST16 / 2; i += 1024) lh[i] = 0;
    __syncthreads();
    for (int64_t p = p0 + threadIdx.x; p < p1; p += 1024) {
        const int64_t y = p / w, x = p - y * w;
        const uint32_t v = img[y * pitch + x];
        atomicAdd(&lh[v >> 1], (v & 1) ? 0x10000u : 1u);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < HIST16 / 2; i += 1024) {
        const uint32_t c = lh[i];
        if (c & 0xffffu) atomicAdd(&hist[2 * i], c & 0xffffu);
        if (c >> 16) atomicAdd(&hist[2 * i + 1], c >> 16);
    }
}

__global__ __launch_bounds__(256) void k_image_hist8(const uint8_t* __restrict__ img, int64_t h, int64_t w, int64_t pitch,
                                                     uint32_t* __restrict__ hist) {
    __shared__ uint32_t lh[256];
    const int64_t n = h * w;
    lh[threadIdx.x] = 0;
    __syncthreads();
    for (int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x; p < n; p += (int64_t)gridDim.x * 256) {
        const int64_t y = p / w, x = p - y * w;
        atomicAdd(&lh[img[y * pitch + x]], 1u);
    }
    __syncthreads();
    if (lh[threadIdx.x]) atomicAdd(&hist[threadIdx.x], lh[threadIdx.x]);
}

// ---- exact order statistics of a uint16 image (np.percentile / np.max, solex_util.py:535-537) ---------------
// MSB-first radix select on the 16-bit values: pass 0 histograms the high byte, pass 1 the low byte of the
// pixels whose high byte was chosen.  hist: [n_ranks][2][256] u32, zeroed.  Every workgroup replays pass 0's
// choice with a workgroup-wide scan (one bin per thread).
template <int SLOTS = SEL_SLOTS>                             // (a compile-time count: the copies' loads are issued together, not one after the other)
__device__ __forceinline__ void pick_digit(const uint32_t* __restrict__ hist, int slot_stride, int64_t rank, int& digit, int64_t& below) {
    __shared__ int64_t wave_tot[16];
    __shared__ int64_t chosen[2];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int64_t c = 0;                                         // one bin per thread; a wider workgroup's other threads idle
    if (tid < 256) {
        uint32_t v[SLOTS];
#pragma unroll
        for (int k = 0; k < SLOTS; ++k) v[k] = hist[(int64_t)k * slot_stride + tid];
#pragma unroll
        for (int k = 0; k < SLOTS; ++k) c += v[k];
    }
    int64_t incl = c;
    incl = shg::wave_scan(incl);
    if (lane == 63) wave_tot[wave] = incl;
    __syncthreads();
    for (int i = 0; i < wave; ++i) incl += wave_tot[i];
    const int64_t excl = incl - c;
    if (excl <= rank && rank < incl) { chosen[0] = tid; chosen[1] = excl; }
    __syncthreads();
    digit = (int)chosen[0];
    below = chosen[1];
    __syncthreads();
}

// grid (row chunks), 256 threads.  hist layout: [0][256] = high-byte histogram shared by all ranks,
// [1 + r][256] = low-byte histogram of rank r.  The image is read once per pass, whatever the number of ranks.
// The pass is bound by the LDS atomic rate (about one lane-atomic per clock per CU): what helps is spreading the
// work over every CU (~8192 pixels per workgroup) and keeping same-address collisions short -- a solar frame puts
// most pixels of a wave into a handful of bins, so each bin is kept in interleaved copies ([bin][copy]: a bin's
// copies sit in different LDS banks; 16 copies in pass 0, 4 per rank in pass 1).  Tried and dropped: merging runs
// of equal keys inside a lane's eight pixels (the divergent bookkeeping costs more than the atomics it saves).
constexpr int SEL_COPIES0 = 16, SEL_COPIES1 = 4;


struct SelectPassArgs {
    shg::PtrBatch imgs;
    int64_t h, w, pitch;
    int pass;
    Ranks8 ranks;
    int n_ranks;
    uint32_t* hist;
    int vec_ok;
    size_t zs;
    int sel_window;                  // pass 1: ranks inside the blend kernel's window (and the maximum) need no second pass
    int64_t n_px;
};

__global__ __launch_bounds__(1024) void k_select16_pass(const SelectPassArgs kargs) {
    const shg::PtrBatch& imgs = kargs.imgs;
    const int64_t h = kargs.h, w = kargs.w, pitch = kargs.pitch;
    const int pass = kargs.pass, n_ranks = kargs.n_ranks, vec_ok = kargs.vec_ok;
    const Ranks8& ranks = kargs.ranks;
    uint32_t* __restrict__ hist = kargs.hist;
    const size_t zs = kargs.zs;
    __shared__ uint32_t lh[8 * 256 * SEL_COPIES1];        // pass 0: [bin][16 copies]; pass 1: [rank][bin][4 copies]
    const uint16_t* __restrict__ img = imgs.at<const uint16_t>(blockIdx.z);
    hist = zdisk(hist, zs, blockIdx.z);
    __shared__ int his_s[8];
    if (pass == 1) {
        // replay pass 0's choice for every rank: ONE scan of the shared high-byte histogram (a scan per rank was a third of
        // this kernel's 15 us: every workgroup pays it before its first pixel)
        __shared__ int64_t wave_tot[16];
        const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
        const int stride = (1 + n_ranks) * 256;
        int64_t c = 0;
        if (tid < 256) {
            uint32_t v[SEL_SLOTS];
#pragma unroll
            for (int k = 0; k < SEL_SLOTS; ++k) v[k] = hist[(int64_t)k * stride + tid];
#pragma unroll
            for (int k = 0; k < SEL_SLOTS; ++k) c += v[k];
        }
        int64_t incl = c;
        incl = shg::wave_scan(incl);
        if (lane == 63) wave_tot[wave] = incl;
        __syncthreads();
        for (int i = 0; i < wave; ++i) incl += wave_tot[i];
        const int64_t excl = incl - c;
        for (int r = 0; r < n_ranks; ++r)
            if (excl <= ranks.v[r] && ranks.v[r] < incl) his_s[r] = tid;
    }
    const int n_words = pass == 0 ? 256 * SEL_COPIES0 : n_ranks * 256 * SEL_COPIES1;
    const int nt = blockDim.x;
    for (int i = threadIdx.x; i < n_words; i += nt) lh[i] = 0;
    __syncthreads();
    int his[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) his[r] = (pass == 1 && r < n_ranks) ? his_s[r] : -1;
    if (pass == 1 && kargs.sel_window) {
        const uint32_t* win = hist + (int64_t)SEL_SLOTS * (1 + n_ranks) * 256;
        const uint32_t w0 = sel_window_origin(reinterpret_cast<const SelWin*>(win + SEL_WIN_WORDS));
        bool covered = true;
        for (int r = 0; r < n_ranks; ++r)
            covered = covered && (ranks.v[r] == kargs.n_px - 1 || (uint32_t)(his[r] - (int)w0) < (uint32_t)SEL_NW);
        if (covered) return;                                 // (the same for every workgroup of the disk: they replayed the same histogram)
    }
    // the ranks' high bytes, four to a word (a spare byte repeats the first rank's)
    uint32_t wanted0 = 0, wanted1 = 0;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        wanted0 |= (uint32_t)((r < n_ranks ? his[r] : his[0]) & 0xff) << (8 * r);
        wanted1 |= (uint32_t)((4 + r < n_ranks ? his[4 + r] : his[0]) & 0xff) << (8 * r);
    }
    const int copy0 = threadIdx.x & (SEL_COPIES0 - 1), copy1 = threadIdx.x & (SEL_COPIES1 - 1);
    auto count = [&](uint32_t v) {
        if (pass == 0) {
            atomicAdd(&lh[(v >> 8) * SEL_COPIES0 + copy0], 1u);
        } else {
            const int hi = (int)(v >> 8);
#pragma unroll
            for (int r = 0; r < 8; ++r)
                if (hi == his[r]) atomicAdd(&lh[(r * 256 + (v & 0xff)) * SEL_COPIES1 + copy1], 1u);
        }
    };
    const int64_t rows_per_block = (h + gridDim.x - 1) / gridDim.x;
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
    const int64_t nrows = min(h, r0 + rows_per_block) - r0;
    if (nrows > 0) {
        const int64_t vpr = vec_ok ? w / 8 : 0;           // 16-byte vectors per row
        // four independent 16-byte loads per lane before the first use; out-of-range slots re-read vector 0 and are dropped
        const int64_t nvec = nrows * vpr;
        for (int64_t base = 0; base < nvec; base += 4 * nt) {
            uint4 q[4];
            bool ok[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int64_t i = base + u * nt + threadIdx.x;
                ok[u] = i < nvec;
                const uint32_t ii = ok[u] ? (uint32_t)i : 0u;        // (a workgroup's share is far below 2^32 vectors:
                const uint32_t r = ii / (uint32_t)vpr, vx = ii - r * (uint32_t)vpr;      //  32-bit division, a fifth of the 64-bit one)
                q[u] = *reinterpret_cast<const uint4*>(img + (r0 + r) * pitch + vx * 8);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (!ok[u]) continue;
                const uint32_t d[4] = {q[u].x, q[u].y, q[u].z, q[u].w};
                if (pass == 1) {
                    // Pass 1 counts only the pixels whose high byte is one of the ranks': a fraction of a percent each.  A pair of
                    // pixels is tested without a branch -- is a byte of `wanted` (the ranks' high bytes, four to a word) equal to
                    // a pixel's high byte: the zero-byte test on wanted ^ (byte * 0x01010101) -- and only a pair that has one goes
                    // through count().  (A compare and a branch per pixel and rank made the pass issue bound: 2100 scalar and 1500
                    // vector instructions per wave; a wave always holds some lane with a match, so testing whole vectors first
                    // saves nothing.)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        uint32_t any = 0;
#pragma unroll
                        for (int e = 0; e < 2; ++e) {
                            const uint32_t rep = ((d[j] >> (8 + 16 * e)) & 0xffu) * 0x01010101u;
                            const uint32_t x0 = wanted0 ^ rep, x1 = wanted1 ^ rep;
                            any |= ((x0 - 0x01010101u) & ~x0) | ((x1 - 0x01010101u) & ~x1);
                        }
                        if (any & 0x80808080u) {
                            count(d[j] & 0xffffu);
                            count(d[j] >> 16);
                        }
                    }
                    continue;
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    count(d[j] & 0xffffu);
                    count(d[j] >> 16);
                }
            }
        }
        const int64_t tail = w - vpr * 8;
        for (int64_t i = threadIdx.x; i < nrows * tail; i += nt) {
            const int64_t r = i / tail, x = vpr * 8 + (i - r * tail);
            count(img[(r0 + r) * pitch + x]);
        }
    }
    __syncthreads();
    if (threadIdx.x >= 256) return;
    hist += (int64_t)(blockIdx.x % SEL_SLOTS) * (1 + n_ranks) * 256;
    if (pass == 0) {
        uint32_t c = 0;
#pragma unroll
        for (int k = 0; k < SEL_COPIES0; ++k) c += lh[threadIdx.x * SEL_COPIES0 + k];
        if (c) atomicAdd(&hist[threadIdx.x], c);
    } else {
        for (int r = 0; r < n_ranks; ++r) {
            uint32_t c = 0;
#pragma unroll
            for (int k = 0; k < SEL_COPIES1; ++k) c += lh[(r * 256 + threadIdx.x) * SEL_COPIES1 + k];
            if (c) atomicAdd(&hist[(1 + r) * 256 + threadIdx.x], c);
        }
    }
}

// grid (n_ranks), 256 threads
struct SelectFinalArgs {
    Ranks8 ranks;
    uint32_t* hist;
    double* out;
    size_t zs;
    int out_zstride;
    int sel_window;
    int64_t n_px;
};

__global__ __launch_bounds__(256) void k_select16_final(const SelectFinalArgs kargs) {
    const Ranks8& ranks = kargs.ranks;
    uint32_t* __restrict__ hist = kargs.hist;
    double* __restrict__ out = kargs.out;
    const size_t zs = kargs.zs;
    const int out_zstride = kargs.out_zstride;
    hist = zdisk(hist, zs, blockIdx.z);
    out += (int64_t)blockIdx.z * out_zstride;
    const int n_ranks = (int)gridDim.x, stride = (1 + n_ranks) * 256;
    int hi, lo;
    int64_t below, below2;
    pick_digit(hist, stride, ranks.v[blockIdx.x], hi, below);
    if (kargs.sel_window) {
        uint32_t* win = hist + (int64_t)SEL_SLOTS * stride;
        SelWin* st = reinterpret_cast<SelWin*>(win + SEL_WIN_WORDS);
        const uint32_t w0 = sel_window_origin(st);
        if (blockIdx.x == 0 && threadIdx.x == 0)            // the first rank's high byte in the middle of the next scan's window
            st->w0_next = (uint32_t)min(max(hi - SEL_NW / 2 + 1, 0), 256 - SEL_NW);
        if (ranks.v[blockIdx.x] == kargs.n_px - 1) {         // np.max: the blend kernel's atomic maximum
            if (threadIdx.x == 0) out[blockIdx.x] = (double)__hip_atomic_load(&st->max, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return;
        }
        const uint32_t d = (uint32_t)(hi - (int)w0);
        if (d < (uint32_t)SEL_NW) {                          // the blend kernel has counted this high byte's low bytes (always valid)
            pick_digit<SEL_WIN_SLOTS>(win + d * 256, SEL_NW * 256, ranks.v[blockIdx.x] - below, lo, below2);
            if (threadIdx.x == 0) out[blockIdx.x] = (double)((hi << 8) | lo);
            return;
        }
    }
    pick_digit(hist + (1 + blockIdx.x) * 256, stride, ranks.v[blockIdx.x] - below, lo, below2);
    if (threadIdx.x == 0) out[blockIdx.x] = (double)((hi << 8) | lo);
}

// Totals of the summed tile histograms over runs of 64 bins: chunk[c] = sum over tiles and bins 64c .. 64c+63.
// grid 64 x 1024 threads: a wave's 64 lanes hold exactly one run.
__global__ __launch_bounds__(1024) void k_chunk_sums(const uint32_t* __restrict__ hist, int ntiles, uint32_t* __restrict__ chunk) {
    const int bin = blockIdx.x * 1024 + threadIdx.x;
    uint32_t t = 0;
    for (int k = 0; k < ntiles; ++k) t += hist[(int64_t)k * HIST16 + bin];
    t = shg::wave_sum(t);
    if ((threadIdx.x & 63) == 0) chunk[bin >> 6] = t;
}

// Order statistics of the image whose per-tile histograms CLAHE has just built (valid when the tile grid divides the
// image: no reflected padding in the histograms).  One workgroup per rank: lane t takes the t-th run of 64 bins from the
// chunk sums of k_chunk_sums, a workgroup scan finds the run that holds the rank, one wave scans its 64 bins.
struct HistRanksArgs {
    const uint32_t *hist, *chunk_sums;
    int chunk_sets, ntiles;
    Ranks8 ranks;
    double* out;
    size_t zs;
    int out_zstride;
};

__global__ __launch_bounds__(1024) void k_hist_ranks(const HistRanksArgs kargs) {
    const uint32_t* __restrict__ hist = kargs.hist;
    const uint32_t* __restrict__ chunk_sums = kargs.chunk_sums;
    const int chunk_sets = kargs.chunk_sets, ntiles = kargs.ntiles, out_zstride = kargs.out_zstride;
    const Ranks8& ranks = kargs.ranks;
    double* __restrict__ out = kargs.out;
    const size_t zs = kargs.zs;
    hist = zdisk(hist, zs, blockIdx.z);
    chunk_sums = zdisk(chunk_sums, zs, blockIdx.z);
    out += (int64_t)blockIdx.z * out_zstride;
    hist_rank_job(hist, chunk_sums, chunk_sets, ntiles, ranks.v[blockIdx.x], out + blockIdx.x);
}

void ensure_lds_attr() {
    static const bool done = [] {                        // (a function-local static: once, also with several pool threads here)
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_tile_hist16), hipFuncAttributeMaxDynamicSharedMemorySize, HIST16 * 2);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_tile_hist16_slices<false, 16>), hipFuncAttributeMaxDynamicSharedMemorySize, HIST16 * 2);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_tile_hist16_slices<false, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, HIST16 * 2);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_tile_hist16_slices<false, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, HIST16 * 2);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_tile_hist16_slices<true, 16>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  HIST16 * 2 + kFusedMaxSliceRows * 8);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_tile_hist16_slices<true, 8>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  HIST16 * 2 + kFusedMaxSliceRows * 8);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_tile_hist16_slices<true, 4>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  HIST16 * 2 + kFusedMaxSliceRows * 8);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_tile_lut16_blocks<true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 2048 * 16 * 2);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_tile_lut16_blocks<false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 2048 * 16 * 2);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_image_hist16), hipFuncAttributeMaxDynamicSharedMemorySize, HIST16 * 2);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_tile_lut16_lds), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (HIST16 + HIST16 / 64) * 2);
        return true;
    }();
    (void)done;
}

}  // namespace

extern "C" size_t shg_clahe_workspace_bytes(int tiles, int bytes_per_px) {
    if (tiles < 1 || tiles > 16 || (bytes_per_px != 1 && bytes_per_px != 2)) return 0;
    const size_t hist = bytes_per_px == 1 ? 256 : HIST16;
    return (size_t)tiles * tiles * hist * (sizeof(uint32_t) + sizeof(uint16_t));
}

namespace {
// sel_hist (may be NULL): the zeroed slot histograms of the select that follows on dst; the kernel then counts its first pass
// The disks of one launch: their images, where their results go, how many, and the byte distance between their workspaces.
struct RanksJob {                                        // two order statistics of every disk's frame -> out[disk * out_zstride + {0, 1}]
    int64_t rank[2];
    double* out;
    int out_zstride;
};
struct Disks {
    shg::PtrBatch src, dst;
    int n;
    size_t zs;
    bool fused = false;              // src is still to be made, from `from` (k_tile_hist16_slices<true>)
    FusedSrc from = {};
    bool aligned(unsigned mask) const {
        uintptr_t bits = 0;
        for (int i = 0; i < n; ++i) bits |= reinterpret_cast<uintptr_t>(src.p[i]) | reinterpret_cast<uintptr_t>(dst.p[i]);
        return (bits & mask) == 0;
    }
};
inline Disks one_disk(const void* src, void* dst) {
    Disks d;
    d.src = {};
    d.dst = {};
    d.src.p[0] = src;
    d.dst.p[0] = dst ? dst : src;
    d.n = 1;
    d.zs = 0;
    return d;
}

inline int launch_interp16(const Disks& d, int64_t h, int64_t w, int64_t pitch, int tiles, float inv_tw, float inv_th,
                           const uint16_t* lut, bool value_major, int64_t dst_pitch, uint32_t* sel_hist, int sel_stride,
                           hipStream_t st, bool* counted, bool sel_window = false) {
    *counted = false;
    const unsigned nz = (unsigned)d.n;
    if (!value_major) {                                  // (single image only: the caller checked)
        k_clahe_interp<uint16_t, HIST16><<<dim3((unsigned)((w + 255) / 256), (unsigned)h), 256, 0, st>>>(
            static_cast<const uint16_t*>(d.src.p[0]), h, w, pitch, tiles, inv_tw, inv_th, lut, static_cast<uint16_t*>(const_cast<void*>(d.dst.p[0])), dst_pitch);
        return shg::check_launch("k_clahe_interp");
    }
    const bool vec = d.aligned(7) && pitch % 4 == 0 && dst_pitch % 4 == 0;
    auto blocks = [&](int px, int rounds) {              // workgroups for the flat (row, vector) sequence
        const int64_t lanes = ((w + px - 1) / px) * h;
        return (unsigned)((lanes + 256 * (int64_t)rounds - 1) / (256 * (int64_t)rounds));
    };
    static const bool tiled_ok = [] { const char* v = getenv("SHG_INTERP_TILED"); return !(v && v[0] == '0'); }();
    InterpVmArgs a{d.src, h, w, pitch, tiles, inv_tw, inv_th, lut, d.dst, dst_pitch, 1, nullptr, 0, d.zs, 0, 1u, 0};
    if (vec && sel_hist) {
        const int rows = 4;                              // rounds per workgroup: amortises the histogram's zeroing and flush
        a.rows = rows;
        a.sel_hist = sel_hist;
        a.sel_stride = sel_stride;
        a.sel_window = sel_window ? 1 : 0;
        *counted = true;
        if (tiled_ok) {
            // 4 lanes x 16 rows per wave, 4 waves across: 64 pixels x 16 rows a round (measured over 21 disks: 243 us; 2 waves
            // across 242, one 270; 2 lanes x 32 rows 262-384; 8 lanes x 8 rows 246-253; the flat sequence 300)
            // Eight pixels a lane (16-byte loads and stores) from four disks up: 222 us against 243 over 21 disks -- and 17.5 against 13.7 us
            // on one, where the launch is too small to fill the device.  (SHG_INTERP_SHAPE = lw | wx << 4 | (8 px) << 8: tools' sweeps;
            // halving the kernel's L2 requests this way does not change what it costs a pass A beside it, profiles/r04_sweeps.txt.)
            const char* shape_env = getenv("SHG_INTERP_SHAPE");      // (read per call: a test compares the two lane widths in one process)
            const int shape = shape_env ? atoi(shape_env) : 0;
            const int lw = shape ? (shape & 15) : 2, wx = shape ? ((shape >> 4) & 15) : 2;
            const bool px8 = (shape ? ((shape >> 8) & 1) != 0 : nz >= 4) && d.aligned(15) && pitch % 8 == 0 && dst_pitch % 8 == 0;
            const int pxn = px8 ? 8 : 4;
            const int64_t wg_px = (int64_t)pxn << (lw + wx), wg_rows = (int64_t)(64 >> lw) * (4 >> wx);
            const uint32_t tx = (uint32_t)((w + wg_px - 1) / wg_px), ty = (uint32_t)((h + wg_rows * rows - 1) / (wg_rows * rows));
            a.tiled = 0x10000 | lw | (wx << 8);
            a.tiles_x = tx;
            const dim3 g(tx * ty, 1u, nz);
            if (tiles == 2) return px8 ? shg::launch(k_clahe_interp_vm<8, true, true>, g, dim3(256), 0, st, a, "k_clahe_interp_vm")
                                       : shg::launch(k_clahe_interp_vm<4, true, true>, g, dim3(256), 0, st, a, "k_clahe_interp_vm");
            if (px8) return shg::launch(k_clahe_interp_vm<8, true, false>, g, dim3(256), 0, st, a, "k_clahe_interp_vm");
            return shg::launch(k_clahe_interp_vm<4, true, false>, g, dim3(256), 0, st, a, "k_clahe_interp_vm");
        }
        if (tiles == 2) return shg::launch(k_clahe_interp_vm<4, true, true>, dim3(blocks(4, rows), 1u, nz), dim3(256), 0, st, a, "k_clahe_interp_vm");
        return shg::launch(k_clahe_interp_vm<4, true, false>, dim3(blocks(4, rows), 1u, nz), dim3(256), 0, st, a, "k_clahe_interp_vm");
    }
    if (vec) return tiles == 2 ? shg::launch(k_clahe_interp_vm<4, false, true>, dim3(blocks(4, 1), 1u, nz), dim3(256), 0, st, a, "k_clahe_interp_vm")
                               : shg::launch(k_clahe_interp_vm<4, false, false>, dim3(blocks(4, 1), 1u, nz), dim3(256), 0, st, a, "k_clahe_interp_vm");
    return tiles == 2 ? shg::launch(k_clahe_interp_vm<1, false, true>, dim3(blocks(1, 1), 1u, nz), dim3(256), 0, st, a, "k_clahe_interp_vm")
                      : shg::launch(k_clahe_interp_vm<1, false, false>, dim3(blocks(1, 1), 1u, nz), dim3(256), 0, st, a, "k_clahe_interp_vm");
}

// tile geometry as OpenCV pads it (copyMakeBorder(0, t - h%t, 0, t - w%t, REFLECT_101), clahe.cpp)
inline void tile_geometry(int64_t h, int64_t w, int tiles, int64_t* th, int64_t* tw) {
    int64_t he = h, we = w;
    if (!(w % tiles == 0 && h % tiles == 0)) {
        he = h + (tiles - h % tiles);
        we = w + (tiles - w % tiles);
    }
    *th = he / tiles;
    *tw = we / tiles;
}

// a tile's slices are runs of whole rows holding at most slice_px pixels (a tile row longer than that: one row a slice --
// the u16 counters of such a slice can wrap, and clahe_impl keeps those images off this path)
inline int64_t slice_rows_of(int64_t tw, int64_t slice_px) { return tw >= slice_px ? 1 : slice_px / tw; }
inline int64_t slice_count(int64_t th, int64_t tw, int64_t slice_px) {
    const int64_t rows = slice_rows_of(tw, slice_px);
    return (th + rows - 1) / rows;
}

// extra workspace of the atomics-free 16-bit path, after the [hist | lut] block: slice histograms, chunk sums, se
struct FastLayout { size_t part, chunk, se, total; int64_t slices; };
inline FastLayout fast_layout(int64_t h, int64_t w, int tiles) {
    int64_t th, tw;
    tile_geometry(h, w, tiles, &th, &tw);
    FastLayout f;
    const size_t ntiles = (size_t)tiles * tiles;
    f.slices = slice_count(th, tw, SLICE_PX);
    f.part = 0;
    f.chunk = ntiles * (size_t)f.slices * (HIST16 / 2) * sizeof(uint32_t);
    f.se = f.chunk + ntiles * 1024 * sizeof(uint32_t);
    f.total = f.se + ntiles * 128 * sizeof(int32_t);         // (32 x 2 for the u16 slices, 128 for the saturated ones)
    return f;
}
}  // namespace

extern "C" size_t shg_clahe_workspace_bytes_for(int64_t h, int64_t w, int tiles, int bytes_per_px) {
    const size_t base = shg_clahe_workspace_bytes(tiles, bytes_per_px);
    if (base == 0 || h <= 0 || w <= 0) return 0;
    if (bytes_per_px != 2) return base;
    return (base + 255) / 256 * 256 + fast_layout(h, w, tiles).total;
}

namespace {
// chunk_tile_out (may be NULL): where the per-tile 64-bin chunk sums were left, or NULL when the call took the
// histogram-with-atomics path (small workspace, 8-bit image, clip out of the u16 range).
// disks (may be NULL: the one image img -> dst): several images of one shape in one launch per kernel, disk i working in the
// workspace at `workspace + i * disks->zs`; only the atomics-free 16-bit path takes more than one.
int clahe_impl(const void* img, int64_t h, int64_t w, int64_t pitch, int bytes_per_px, double clip_limit, int tiles,
               void* dst, int64_t dst_pitch, void* workspace, size_t workspace_bytes, shg_stream_t stream, const uint32_t** chunk_tile_out,
               uint32_t* sel_hist, int sel_stride, bool* sel_pass0_done, const Disks* disks = nullptr, bool* sel_zeroed = nullptr,
               const RanksJob* ranks_job = nullptr, bool* ranks_done = nullptr, bool* sel_window = nullptr) {
    // sel_window (in / out): the caller's sel_hist has the window's room behind its slots (select_window_bytes) and wants it used;
    // -> false when this call did not take the path that fills it
    const bool want_window = sel_window && *sel_window;
    if (sel_window) *sel_window = false;
    if (ranks_done) *ranks_done = false;
    if (sel_zeroed) *sel_zeroed = false;                 // -> true when the histogram reduction has zeroed sel_hist (the caller asked by passing it)
    if (chunk_tile_out) *chunk_tile_out = nullptr;
    if (sel_pass0_done) *sel_pass0_done = false;
    SHG_REQUIRE(img && dst && workspace, SHG_E_ARG, "shg_clahe: null pointer");
    const Disks dset = disks ? *disks : one_disk(img, dst);
    SHG_REQUIRE(h > 0 && w > 0 && pitch >= w && dst_pitch >= w, SHG_E_ARG, "shg_clahe: bad image size");
    SHG_REQUIRE(bytes_per_px == 1 || bytes_per_px == 2, SHG_E_ARG, "shg_clahe: bytes_per_px must be 1 or 2");
    SHG_REQUIRE(tiles >= 1 && tiles <= 16, SHG_E_UNSUPPORTED, "shg_clahe: tiles must be in 1..16");
    SHG_REQUIRE(h < 65536, SHG_E_UNSUPPORTED, "shg_clahe: more than 65535 rows");
    SHG_REQUIRE(workspace_bytes >= shg_clahe_workspace_bytes(tiles, bytes_per_px), SHG_E_WORKSPACE, "shg_clahe: workspace too small");
    // a reflect-101 extension needs at least `tiles` + 1 pixels along an extended axis
    SHG_REQUIRE((h % tiles == 0 && w % tiles == 0) || (h > tiles && w > tiles), SHG_E_UNSUPPORTED, "shg_clahe: image smaller than the tile grid");
    hipStream_t st = shg::as_stream(stream);
    const int hist_size = bytes_per_px == 1 ? 256 : HIST16;
    const int ntiles = tiles * tiles;
    int64_t th, tw;
    tile_geometry(h, w, tiles, &th, &tw);
    const int64_t area = th * tw;
    SHG_REQUIRE(area < (1ll << 31), SHG_E_UNSUPPORTED, "shg_clahe: tile too large");
    const float lut_scale = (float)(hist_size - 1) / (float)area;
    int clip = 0;
    if (clip_limit > 0.0) {
        clip = (int)(clip_limit * (double)area / hist_size);
        clip = clip > 1 ? clip : 1;
    }
    uint32_t* hist = static_cast<uint32_t*>(workspace);
    uint16_t* lut = reinterpret_cast<uint16_t*>(hist + (size_t)ntiles * hist_size);
    const float inv_tw = 1.0f / (float)tw, inv_th = 1.0f / (float)th;
    dim3 igrid((unsigned)((w + 255) / 256), (unsigned)h);
    if (bytes_per_px == 2 && clip > 0 && clip <= 65535 && tw <= 65535 && workspace_bytes >= shg_clahe_workspace_bytes_for(h, w, tiles, 2)) {
        // no atomics, no memset: slice histograms stored whole, one reduction, the LUT by 32 workgroups per tile
        ensure_lds_attr();
        const FastLayout f = fast_layout(h, w, tiles);
        char* extra = static_cast<char*>(workspace) + (shg_clahe_workspace_bytes(tiles, 2) + 255) / 256 * 256;
        uint32_t* part = reinterpret_cast<uint32_t*>(extra + f.part);
        uint32_t* chunk_tile = reinterpret_cast<uint32_t*>(extra + f.chunk);
        int32_t* se = reinterpret_cast<int32_t*>(extra + f.se);
        bool sat_path = false;
        int64_t sat_k_top[2] = {0, 0};
        { SHG_PROF("clahe_hist", st);
          const unsigned nz = (unsigned)dset.n;
          // One image: 32768-pixel slices spread a tile over enough workgroups to fill the chip.  A stack of disks fills it
          // anyway: the largest slice a u16 counter allows halves the slice histograms written here and read back by the
          // reduction (the layout's f.slices is the upper bound the workspace was sized for).
          static const int64_t big = [] { const char* v = getenv("SHG_CLAHE_SLICE_PX"); return v ? (int64_t)atoi(v) : (int64_t)65535; }();
          static const int big_from = [] { const char* v = getenv("SHG_CLAHE_BIG_FROM"); return v ? atoi(v) : 4; }();      // disks per launch from which the big slices are taken
          // Saturated counters (k_tile_hist16_slices<., 8 / 4>): whenever the clip limit fits a byte and nobody needs the true counts
          // -- the frame's order statistics may ride along when they lie within `clip` pixels of the top (np.percentile(frame, 99.9999)
          // of a 4 Mpx image: the 5th and 6th largest), a caller who wants the chunk sums (shg_contrast_stats_u16) gets the u16 slices.
          // (read at every call, three getenv: the tests hold one setting against another in one process)
          const int sat_mode = [] { const char* v = getenv("SHG_CLAHE_SAT"); return v ? atoi(v) : 1; }();         // 0: never; 8: bytes even where nibbles would do
          const int64_t sat_px = [] { const char* v = getenv("SHG_CLAHE_SAT_PX"); return v && atoi(v) > 0 ? (int64_t)atoi(v) : (int64_t)0; }();      // pixels per slice (0: chosen below)
          const bool want_ranks = ranks_job && ranks_done;
          bool sat = sat_mode != 0 && clip <= 255 && tw <= 65280 && (chunk_tile_out == nullptr || want_ranks);
          int64_t k_top[2] = {0, 0};
          if (sat && want_ranks) {
              for (int r = 0; r < 2; ++r) {
                  k_top[r] = h * w - ranks_job->rank[r];         // rank (from the bottom, 0-based) -> the k-th largest
                  sat = sat && k_top[r] >= 1 && k_top[r] <= clip;
              }
          }
          const int bits = !sat ? 16 : (clip <= 15 && sat_mode != 8 ? 4 : 8);
          int64_t slice_px = (dset.n >= big_from && big > SLICE_PX && big <= 65535) ? big : SLICE_PX;
          if (sat && sat_px) slice_px = sat_px;
          const int64_t chunk_rows = slice_rows_of(tw, 65280);                    // (sat: rows counted between two clamps of the u16 counters)
          int64_t slice_rows = slice_rows_of(tw, slice_px), slices = slice_count(th, tw, slice_px);
          if (sat) {
              // as many slices as the byte sums of k_hist_reduce_sat hold (256) and the workspace has room for (sized for f.slices u16 slices)
              const int64_t max_slices = std::min<int64_t>(256, f.slices * (16 / bits));
              if (!sat_px) {
                  // A slice may be any length now, and a workgroup has a CU to itself (its histogram is 128 KB of the 160 KB of LDS): the
                  // launch takes ceil(workgroups / CUs) rounds of one slice each.  The number of slices per tile that makes rounds x
                  // (pixels of a slice + what zeroing, clamping and storing 64 K counters costs, about 8000 pixels' worth) smallest:
                  // 63 at one 2000 x 2098 disk (252 workgroups, one round), 3 for a stack of 21 (252 again) -- not the 4 that
                  // would leave two thirds of the chip idle in a second round.
                  int64_t best = 1, best_cost = INT64_MAX;
                  for (int64_t sl = 1; sl <= std::min<int64_t>(max_slices, th); ++sl) {
                      const int64_t rows = (th + sl - 1) / sl, n_sl = (th + rows - 1) / rows;
                      if (dset.fused && rows > kFusedMaxSliceRows) continue;
                      const int64_t wgs = n_sl * ntiles * (int64_t)dset.n, rounds = (wgs + shg::kCUs - 1) / shg::kCUs;
                      const int64_t cost = rounds * (rows * tw + 8192);
                      if (cost < best_cost) { best_cost = cost; best = rows; }
                  }
                  slice_rows = best;
              }
              slices = (th + slice_rows - 1) / slice_rows;
              if (slices > max_slices) slice_rows = (th + max_slices - 1) / max_slices;
              if (dset.fused) slice_rows = std::min<int64_t>(slice_rows, kFusedMaxSliceRows);      // (the factors' room in LDS; th < 65536: 32 slices at most)
              slices = (th + slice_rows - 1) / slice_rows;
          }
          // (tw >= 64: the last tile column then holds columns of the image itself, and a row's loose ends fit half a wave)
          int vec = tw >= 64 && pitch % 8 == 0 && dset.aligned(15);
          const dim3 hgrid((unsigned)slices, (unsigned)ntiles, nz);
          if (dset.fused) {
              SHG_REQUIRE(slice_rows <= kFusedMaxSliceRows, SHG_E_ARG, "shg_clahe: the fused histogram's slices hold at most %d rows", kFusedMaxSliceRows);
              uintptr_t abits = 0;
              for (int i = 0; i < dset.n; ++i) abits |= reinterpret_cast<uintptr_t>(dset.from.raw.p[i]);
              vec = vec && (abits & 15) == 0 && dset.from.raw_pitch % 8 == 0 && dset.from.sx0 == 0 && dset.from.dx0 == 0 && dset.from.ncopy == w;
          }
          const HistSlicesArgs ha{dset.src, h, w, pitch, tiles, th, tw, part, dset.zs, (int)slice_rows, vec, dset.fused ? dset.from : FusedSrc{}, clip, (int)chunk_rows};
          const size_t lds = HIST16 * 2 + (dset.fused ? (size_t)slice_rows * 8 : 0);
          int e = 0;
          if (dset.fused) {
              e = bits == 16 ? shg::launch(k_tile_hist16_slices<true, 16>, hgrid, dim3(1024), lds, st, ha, "k_tile_hist16_slices")
                  : bits == 8 ? shg::launch(k_tile_hist16_slices<true, 8>, hgrid, dim3(1024), lds, st, ha, "k_tile_hist16_slices")
                              : shg::launch(k_tile_hist16_slices<true, 4>, hgrid, dim3(1024), lds, st, ha, "k_tile_hist16_slices");
          } else {
              e = bits == 16 ? shg::launch(k_tile_hist16_slices<false, 16>, hgrid, dim3(1024), lds, st, ha, "k_tile_hist16_slices")
                  : bits == 8 ? shg::launch(k_tile_hist16_slices<false, 8>, hgrid, dim3(1024), lds, st, ha, "k_tile_hist16_slices")
                              : shg::launch(k_tile_hist16_slices<false, 4>, hgrid, dim3(1024), lds, st, ha, "k_tile_hist16_slices");
          }
          if (e) return e;
          const bool zero_sel = sel_hist && sel_zeroed;
          const bool window = zero_sel && want_window;
          const HistReduceArgs ra{part, (int)slices, clip, hist, chunk_tile, se, dset.zs, zero_sel ? sel_hist : nullptr, zero_sel ? SEL_SLOTS * sel_stride : 0, window ? 1 : 0};
          e = bits == 16 ? shg::launch(k_hist_reduce, dim3(32, (unsigned)ntiles, nz), dim3(1024), 0, st, ra, "k_hist_reduce")
              : bits == 8 ? shg::launch(k_hist_reduce_sat<8>, dim3(64, (unsigned)ntiles, nz), dim3(256), 0, st, ra, "k_hist_reduce_sat")
                          : shg::launch(k_hist_reduce_sat<4>, dim3(32, (unsigned)ntiles, nz), dim3(256), 0, st, ra, "k_hist_reduce_sat");
          if (e) return e;
          sat_path = sat;
          sat_k_top[0] = k_top[0];
          sat_k_top[1] = k_top[1];
          if (zero_sel) *sel_zeroed = true; }
        { SHG_PROF("clahe_lut", st);
          LutBlocksArgs la{hist, se, clip, lut_scale, lut, dset.zs, 0, {0, 0}, nullptr, 0, nullptr, 0, (int)area, ntiles, BorderPx{{}, h, w, pitch, h, w}};
          if (ranks_job && ranks_done) {                     // the frame's order statistics ride along (two more workgroups per disk)
              la.border = BorderPx{dset.src, h, w, pitch, th * tiles, tw * tiles};
              la.n_ranks = 2;
              la.rank[0] = sat_path ? sat_k_top[0] : ranks_job->rank[0];
              la.rank[1] = sat_path ? sat_k_top[1] : ranks_job->rank[1];
              la.chunk_sums = chunk_tile;
              la.chunk_sets = ntiles;
              la.ranks_out = ranks_job->out;
              la.ranks_zstride = ranks_job->out_zstride;
              *ranks_done = true;
          }
          // (measured: 22.4 against 24.9 us over a 21-disk stack, but 10.6 against 5.0 us for one disk -- the tiles of a block one
          // after the other in 32 workgroups: taken from eight disks a launch on)
          const bool allt = ntiles <= 16 && dset.n >= 8;      // a workgroup builds its 2048 entries for every tile and stores them as one run
          const dim3 lgrid(32u + (unsigned)la.n_ranks, allt ? 1u : (unsigned)ntiles, (unsigned)dset.n);
          const size_t llds = allt ? (size_t)2048 * ntiles * sizeof(uint16_t) : 0;
          int e;
          if (allt) e = sat_path ? shg::launch(k_tile_lut16_blocks<true, true>, lgrid, dim3(1024), llds, st, la, "k_tile_lut16_blocks")
                                 : shg::launch(k_tile_lut16_blocks<false, true>, lgrid, dim3(1024), llds, st, la, "k_tile_lut16_blocks");
          else e = sat_path ? shg::launch(k_tile_lut16_blocks<true, false>, lgrid, dim3(1024), 0, st, la, "k_tile_lut16_blocks")
                            : shg::launch(k_tile_lut16_blocks<false, false>, lgrid, dim3(1024), 0, st, la, "k_tile_lut16_blocks");
          if (e) return e; }
        { SHG_PROF("clahe_interp", st);
          bool counted = false;
          const bool window = sel_hist && sel_zeroed && want_window;
          if (int e = launch_interp16(dset, h, w, pitch, tiles, inv_tw, inv_th, lut, true, dst_pitch, sel_hist, sel_stride, st, &counted, window)) return e;
          if (sel_pass0_done) *sel_pass0_done = counted;
          if (sel_window) *sel_window = window && counted; }
        if (chunk_tile_out) *chunk_tile_out = chunk_tile;
        return 0;
    }
    SHG_REQUIRE(dset.n == 1 && !dset.fused, SHG_E_UNSUPPORTED, "shg_clahe: several disks per launch need the roomy 16-bit workspace");
    if (hipError_t e = hipMemsetAsync(hist, 0, (size_t)ntiles * hist_size * sizeof(uint32_t), st)) {
        shg::set_error("shg_clahe: memset: %s", hipGetErrorString(e));
        return (int)e;
    }
    if (bytes_per_px == 2) {
        ensure_lds_attr();
        dim3 hgrid((unsigned)((area + SLICE_PX - 1) / SLICE_PX), (unsigned)ntiles);
        { SHG_PROF("clahe_hist", st); k_tile_hist16<<<hgrid, 1024, HIST16 * 2, st>>>(static_cast<const uint16_t*>(img), h, w, pitch, tiles, th, tw, hist); }
        if (int e = shg::check_launch("k_tile_hist16")) return e;
        if (clip > 0 && clip <= 65535) {
            SHG_PROF("clahe_lut", st);
            k_tile_lut16_lds<<<ntiles, 1024, (HIST16 + HIST16 / 64) * sizeof(uint16_t), st>>>(hist, clip, lut_scale, lut);
        } else {
            SHG_PROF("clahe_lut", st);
            k_tile_lut<HIST16><<<ntiles, 1024, 0, st>>>(hist, clip, lut_scale, lut);
        }
        if (int e = shg::check_launch("k_tile_lut")) return e;
        { SHG_PROF("clahe_interp", st);
          bool counted = false;
          if (int e = launch_interp16(dset, h, w, pitch, tiles, inv_tw, inv_th, lut, false, dst_pitch, nullptr, 0, st, &counted)) return e; }
    } else {
        int64_t hb = (area + 4095) / 4096;
        if (hb > 256) hb = 256;
        dim3 hgrid((unsigned)hb, (unsigned)ntiles);
        { SHG_PROF("clahe_hist", st); k_tile_hist8<<<hgrid, 256, 0, st>>>(static_cast<const uint8_t*>(img), h, w, pitch, tiles, th, tw, hist); }
        if (int e = shg::check_launch("k_tile_hist8")) return e;
        { SHG_PROF("clahe_lut", st); k_tile_lut<256><<<ntiles, 1024, 0, st>>>(hist, clip, lut_scale, lut); }
        if (int e = shg::check_launch("k_tile_lut")) return e;
        { SHG_PROF("clahe_interp", st); k_clahe_interp<uint8_t, 256><<<igrid, 256, 0, st>>>(static_cast<const uint8_t*>(img), h, w, pitch, tiles, inv_tw, inv_th, lut,
                                                            static_cast<uint8_t*>(dst), dst_pitch); }
    }
    if (int e = shg::check_launch("k_clahe_interp")) return e;
    if (sel_hist && sel_zeroed) {                        // nothing has counted into sel_hist on this path: zero it the plain way
        if (hipError_t e = hipMemsetAsync(sel_hist, 0, (size_t)SEL_SLOTS * sel_stride * sizeof(uint32_t), st)) {
            shg::set_error("shg_clahe: memset: %s", hipGetErrorString(e));
            return (int)e;
        }
        *sel_zeroed = true;
    }
    return 0;
}
}  // namespace

extern "C" int shg_clahe(const void* img, int64_t h, int64_t w, int64_t pitch, int bytes_per_px, double clip_limit, int tiles,
                         void* dst, int64_t dst_pitch, void* workspace, size_t workspace_bytes, shg_stream_t stream) {
    return clahe_impl(img, h, w, pitch, bytes_per_px, clip_limit, tiles, dst, dst_pitch, workspace, workspace_bytes, stream, nullptr, nullptr, 0, nullptr);
}

extern "C" int shg_hist(const void* img, int64_t h, int64_t w, int64_t pitch, int bytes_per_px, uint32_t* hist, shg_stream_t stream) {
    SHG_REQUIRE(img && hist, SHG_E_ARG, "shg_hist: null pointer");
    SHG_REQUIRE(h > 0 && w > 0 && pitch >= w, SHG_E_ARG, "shg_hist: bad image size");
    SHG_REQUIRE(bytes_per_px == 1 || bytes_per_px == 2, SHG_E_ARG, "shg_hist: bytes_per_px must be 1 or 2");
    hipStream_t st = shg::as_stream(stream);
    const int hist_size = bytes_per_px == 1 ? 256 : HIST16;
    if (hipError_t e = hipMemsetAsync(hist, 0, (size_t)hist_size * sizeof(uint32_t), st)) {
        shg::set_error("shg_hist: memset: %s", hipGetErrorString(e));
        return (int)e;
    }
    const int64_t n = h * w;
    if (bytes_per_px == 2) {
        ensure_lds_attr();
        { SHG_PROF("hist", st); k_image_hist16<<<(unsigned)((n + SLICE_PX - 1) / SLICE_PX), 1024, HIST16 * 2, st>>>(static_cast<const uint16_t*>(img), h, w, pitch, hist); }
    } else {
        int64_t hb = (n + 4095) / 4096;
        if (hb > 256) hb = 256;
        { SHG_PROF("hist", st); k_image_hist8<<<(unsigned)hb, 256, 0, st>>>(static_cast<const uint8_t*>(img), h, w, pitch, hist); }
    }
    return shg::check_launch("k_image_hist");
}

extern "C" size_t shg_select_u16_workspace_bytes(int n_ranks) {
    if (n_ranks < 1 || n_ranks > 8) return 0;
    return (size_t)8 * (1 + n_ranks) * 256 * sizeof(uint32_t) + (size_t)n_ranks * sizeof(int64_t);      // SEL_SLOTS histogram copies
}

namespace {
// pass0_done: the slot histograms are zeroed and already hold the high-byte counts (k_clahe_interp_vm<.., true>)
// disks (may be NULL: the one image img): several images per launch, disk i with its histograms at workspace + i * zs and its
// results at out + i * out_zstride
int select_u16_impl(const uint16_t* img, int64_t h, int64_t w, int64_t pitch, const int64_t* host_ranks, int n_ranks,
                    double* out, void* workspace, size_t workspace_bytes, shg_stream_t stream, bool zeroed, bool pass0_done,
                    const Disks* disks = nullptr, int out_zstride = 0, bool window = false) {
    const Disks dset = disks ? *disks : one_disk(img, nullptr);
    SHG_REQUIRE(img && host_ranks && out && workspace, SHG_E_ARG, "shg_select_u16: null pointer");
    SHG_REQUIRE(h > 0 && w > 0 && pitch >= w && n_ranks >= 1 && n_ranks <= 8, SHG_E_ARG, "shg_select_u16: bad sizes");
    SHG_REQUIRE(workspace_bytes >= shg_select_u16_workspace_bytes(n_ranks), SHG_E_WORKSPACE, "shg_select_u16: workspace too small");
    for (int i = 0; i < n_ranks; ++i)
        SHG_REQUIRE(host_ranks[i] >= 0 && host_ranks[i] < h * w, SHG_E_ARG, "shg_select_u16: rank %lld outside the image", (long long)host_ranks[i]);
    hipStream_t st = shg::as_stream(stream);
    uint32_t* hist = static_cast<uint32_t*>(workspace);
    Ranks8 ranks = {};
    for (int i = 0; i < n_ranks; ++i) ranks.v[i] = host_ranks[i];
    if (!zeroed) {
        hipError_t e = hipMemsetAsync(hist, 0, (size_t)SEL_SLOTS * (1 + n_ranks) * 256 * sizeof(uint32_t), st);
        if (e != hipSuccess) { shg::set_error("shg_select_u16: %s", hipGetErrorString(e)); return (int)e; }
    }
    // ~8192 pixels per workgroup, at most 1024 workgroups, whole rows each
    // (measured, tools/bench_select.py: 2048 / 4096 / 8192 / 16384 pixels per workgroup -> 92 / 58 / 45 / 44 us for two ranks)
    // Several disks in one launch: the chip is full with ~2048 workgroups in all, and a workgroup's share then grows with the
    // number of disks (the replay, zeroing and flush around the pixel loop are paid per workgroup).
    static const int64_t wg_target = [] { const char* v = getenv("SHG_SELECT_WGS"); return v ? (int64_t)atoi(v) : (int64_t)2048; }();
    int64_t per_wg = 8192;
    if (dset.n > 1 && h * w * dset.n / per_wg > wg_target) per_wg = h * w * dset.n / wg_target;
    int64_t want = (h * w + per_wg - 1) / per_wg;
    want = want < 1 ? 1 : (want > 1024 ? 1024 : want);
    const unsigned blocks = (unsigned)(h < want ? h : want);
    uintptr_t bits = 0;
    for (int i = 0; i < dset.n; ++i) bits |= reinterpret_cast<uintptr_t>(dset.src.p[i]);
    const int vec_ok = ((bits & 15) == 0) && (pitch % 8 == 0);
    SHG_REQUIRE(zeroed || dset.n == 1, SHG_E_ARG, "shg_select_u16: several disks need their histograms zeroed by the caller");
    SHG_PROF("select_u16", st);
    for (int pass = pass0_done ? 1 : 0; pass < 2; ++pass) {
        // 512 threads: the zeroing / replay / flush around the pixel loop is shared by twice the waves (256 / 512 / 1024
        // threads: 44.7 / 40.6 / 40.3 us for two ranks, tools/bench_select.py)
        if (int err = shg::launch(k_select16_pass, dim3(blocks, 1u, (unsigned)dset.n), dim3(512), 0, st,
                                 SelectPassArgs{dset.src, h, w, pitch, pass, ranks, n_ranks, hist, vec_ok, dset.zs, window ? 1 : 0, h * w}, "k_select16_pass"))
            return err;
    }
    return shg::launch(k_select16_final, dim3((unsigned)n_ranks, 1u, (unsigned)dset.n), dim3(256), 0, st, SelectFinalArgs{ranks, hist, out, dset.zs, out_zstride, window ? 1 : 0, h * w}, "k_select16_final");
}
}  // namespace

extern "C" int shg_select_u16(const uint16_t* img, int64_t h, int64_t w, int64_t pitch, const int64_t* host_ranks, int n_ranks,
                              double* out, void* workspace, size_t workspace_bytes, shg_stream_t stream) {
    return select_u16_impl(img, h, w, pitch, host_ranks, n_ranks, out, workspace, workspace_bytes, stream, false, false);
}

// ---- image_process in two calls (solex_util.py:527-547) -------------------------------------------------------------
// The contrast stage is a dozen small launches around one host decision (three percentiles -> six rescale bounds).
// Two composite entry points issue them from C instead of from a dozen ctypes calls: same kernels, same order.
// the 3-rank select's area of the contrast stage: its slot histograms, then the window's counts and SelWin
static size_t select3_area_bytes() {
    return (size_t)SEL_SLOTS * (1 + 3) * 256 * sizeof(uint32_t) + (size_t)SEL_WIN_WORDS * sizeof(uint32_t) + sizeof(SelWin) + 3 * sizeof(int64_t);
}
// Measured (tools/ab_tables.sh): over a 21-disk stack the second pass falls from 73 to 14 us and the blend kernel grows from 212 to
// 240 us (its workgroups flush their window counts with global atomics) -- 30 us gained; for ONE disk 12.7 -> 6.3 us against
// 12.9 -> 24.8 us -- 6 us lost.  So: from four disks a launch on.  SHG_SELECT_WINDOW=0 / 1: never / always (the tests).
static bool select_window_wanted(int disks) {                // (read at every call: tests hold one setting against the other)
    const char* v = getenv("SHG_SELECT_WINDOW");
    if (v && v[0] == '0') return false;
    if (v && v[0] == '1') return true;
    return disks >= 4;
}

extern "C" size_t shg_contrast_stats_workspace_bytes(int tiles) {
    const size_t c = shg_clahe_workspace_bytes(tiles, 2), s2 = shg_select_u16_workspace_bytes(2), s3 = select3_area_bytes();
    if (c == 0) return 0;
    const size_t chunks = 1024 * sizeof(uint32_t);           // 64-bin chunk sums of the summed tile histograms
    return ((c + 255) / 256 + (s2 + 255) / 256 + (s3 + 255) / 256 + (chunks + 255) / 256) * 256;
}

extern "C" size_t shg_contrast_stats_workspace_bytes_for(int64_t h, int64_t w, int tiles) {
    const size_t base = shg_contrast_stats_workspace_bytes(tiles);
    if (base == 0 || h <= 0 || w <= 0) return 0;
    const size_t c_old = (shg_clahe_workspace_bytes(tiles, 2) + 255) / 256 * 256, c_new = (shg_clahe_workspace_bytes_for(h, w, tiles, 2) + 255) / 256 * 256;
    return base - c_old + c_new;
}

extern "C" int shg_contrast_stats_u16(const uint16_t* frame, int64_t h, int64_t w, int64_t pitch, double clip_limit, int tiles,
                                      uint16_t* cl1, int64_t cl1_pitch, const int64_t* ranks_frame2, const int64_t* ranks_cl13,
                                      double* out5, void* workspace, size_t workspace_bytes, shg_stream_t stream) {
    SHG_REQUIRE(frame && cl1 && ranks_frame2 && ranks_cl13 && out5 && workspace, SHG_E_ARG, "shg_contrast_stats_u16: null pointer");
    const size_t c_old = (shg_clahe_workspace_bytes(tiles, 2) + 255) / 256 * 256, s2 = (shg_select_u16_workspace_bytes(2) + 255) / 256 * 256;
    SHG_REQUIRE(c_old != 0 && workspace_bytes >= shg_contrast_stats_workspace_bytes(tiles), SHG_E_WORKSPACE,
                "shg_contrast_stats_u16: workspace too small or bad tile count");
    SHG_REQUIRE(h > 0 && w > 0, SHG_E_ARG, "shg_contrast_stats_u16: empty image");
    // a workspace of shg_contrast_stats_workspace_bytes_for(h, w, tiles) lets CLAHE build its histograms without atomics
    const bool roomy = workspace_bytes >= shg_contrast_stats_workspace_bytes_for(h, w, tiles);
    const size_t c = roomy ? (shg_clahe_workspace_bytes_for(h, w, tiles, 2) + 255) / 256 * 256 : c_old;
    char* ws = static_cast<char*>(workspace);
    const uint32_t* chunk_tile = nullptr;
    // When the tile grid divides the image, CLAHE's tile histograms (still at the head of its workspace) add up to the
    // histogram of the frame: np.percentile(frame, q)'s two order statistics are read off them (k_chunk_sums, k_hist_ranks)
    // instead of selecting over the image again (two passes of k_select16_pass).
    // the select on the CLAHE image (3 ranks): its histograms are zero before the interpolation kernel, which counts
    // the first pass while the pixels are in its registers
    uint32_t* sel3 = reinterpret_cast<uint32_t*>(ws + c + s2);
    bool pass0_done = false, sel_zeroed = false;         // (zeroed by CLAHE's histogram reduction on the way, or by a memset where that does not run)
    bool window = select_window_wanted(1);
    if (int e = clahe_impl(frame, h, w, pitch, 2, clip_limit, tiles, cl1, cl1_pitch, ws, c, stream, &chunk_tile, sel3, (1 + 3) * 256, &pass0_done, nullptr,
                           &sel_zeroed, nullptr, nullptr, &window))
        return e;
    SHG_REQUIRE(sel_zeroed, SHG_E_RUNTIME, "shg_contrast_stats_u16: the select histograms were not zeroed");
    if (h % tiles == 0 && w % tiles == 0) {
        const size_t s3r = (select3_area_bytes() + 255) / 256 * 256;
        uint32_t* chunk_sums = reinterpret_cast<uint32_t*>(ws + c + s2 + s3r);
        Ranks8 ranks = {};
        for (int i = 0; i < 2; ++i) {
            SHG_REQUIRE(ranks_frame2[i] >= 0 && ranks_frame2[i] < h * w, SHG_E_ARG, "shg_contrast_stats_u16: rank %lld outside the image", (long long)ranks_frame2[i]);
            ranks.v[i] = ranks_frame2[i];
        }
        hipStream_t st = shg::as_stream(stream);
        SHG_PROF("hist_ranks", st);
        if (chunk_tile) {                                // left by k_hist_reduce, one set per tile
            if (int e = shg::launch(k_hist_ranks, dim3(2), dim3(1024), 0, st,
                                   HistRanksArgs{reinterpret_cast<const uint32_t*>(ws), chunk_tile, tiles * tiles, tiles * tiles, ranks, out5, 0, 0}, "k_hist_ranks"))
                return e;
        } else {
            k_chunk_sums<<<64, 1024, 0, st>>>(reinterpret_cast<const uint32_t*>(ws), tiles * tiles, chunk_sums);
            if (int e = shg::check_launch("k_chunk_sums")) return e;
            if (int e = shg::launch(k_hist_ranks, dim3(2), dim3(1024), 0, st,
                                   HistRanksArgs{reinterpret_cast<const uint32_t*>(ws), chunk_sums, 1, tiles * tiles, ranks, out5, 0, 0}, "k_hist_ranks"))
                return e;
        }
    } else if (int e = shg_select_u16(frame, h, w, pitch, ranks_frame2, 2, out5, ws + c, s2, stream)) return e;
    return select_u16_impl(cl1, h, w, cl1_pitch, ranks_cl13, 3, out5 + 2, sel3, shg_select_u16_workspace_bytes(3), stream, true, pass0_done, nullptr, 0,
                           window && pass0_done);
}

// image_process's CLAHE + order statistics for the k disks of a file in one launch per kernel (shg_stage_process_frames; a
// Doppler stack, Solex_recon.py:105-133).  host_frames / host_cl1: device pointers of k images of one shape; out5: [k][5];
// workspace: k areas of shg_contrast_stats_workspace_bytes_for(h, w, tiles) bytes.  The batched launches need the atomics-free
// CLAHE path (clip limit within the u16 range); anything else goes disk by disk through shg_contrast_stats_u16 -- same results
// either way.  The frame's percentiles are read off the tile histograms -- on a grid that does not divide the image less the pixels
// of the reflected border (BorderPx), and there out5[5 i + 0 / 1] may come back NaN: "select over frame i instead" (hist_rank_top_job).
// Whether contrast_stats_batch takes its batched route (one launch per kernel for all disks, the slice histograms, the percentiles
// read off them) for images of this shape -- the route that can also MAKE the images on its way (FrameSource).
// SHG_CONTRAST_BATCH=0 (the tests): every disk through shg_contrast_stats_u16, one after the other -- the route these launches replaced
static bool contrast_batch_allowed() {                       // (read at every call: the tests hold one setting against the other)
    const char* v = getenv("SHG_CONTRAST_BATCH");
    return !(v && v[0] == '0');
}

bool shg::contrast_stats_batches(int64_t h, int64_t w, int tiles, double clip_limit) {
    if (!contrast_batch_allowed()) return false;
    if (!(tiles >= 1 && tiles <= 16 && h > 0 && w > 0 && h < 65536 && clip_limit > 0.0)) return false;
    if (!(h % tiles == 0 && w % tiles == 0) && !(h > tiles && w > tiles)) return false;     // (a reflected border needs tiles + 1 pixels to mirror)
    int64_t th, tw;
    tile_geometry(h, w, tiles, &th, &tw);
    if (th * tw >= (1ll << 31) || tw > 65535) return false;
    const int clip = (int)(clip_limit * (double)(th * tw) / HIST16);
    return clip <= 65535 && slice_rows_of(tw, 65535) <= kFusedMaxSliceRows;
}

int shg::contrast_stats_batch(const uint16_t* const* host_frames, int64_t k, int64_t h, int64_t w, int64_t pitch, double clip_limit, int tiles,
                              uint16_t* const* host_cl1, int64_t cl1_pitch, const int64_t* ranks_frame2, const int64_t* ranks_cl13, double* out5,
                              void* workspace, size_t workspace_bytes, shg_stream_t stream, const FrameSource* from) {
    SHG_REQUIRE(host_frames && host_cl1 && ranks_frame2 && ranks_cl13 && out5 && workspace && k > 0, SHG_E_ARG, "shg_contrast_stats_u16: null pointer");
    SHG_REQUIRE(h > 0 && w > 0, SHG_E_ARG, "shg_contrast_stats_u16: empty image");
    const size_t per = shg_contrast_stats_workspace_bytes_for(h, w, tiles);
    SHG_REQUIRE(per != 0, SHG_E_WORKSPACE, "shg_contrast_stats_u16: bad tile count");
    bool batched = contrast_batch_allowed() && (k > 1 || from) && workspace_bytes >= (size_t)k * per && tiles >= 1 && tiles <= 16 && clip_limit > 0.0 &&
                   ((h % tiles == 0 && w % tiles == 0) || (h > tiles && w > tiles));
    if (batched) {
        int64_t th, tw;
        tile_geometry(h, w, tiles, &th, &tw);
        const int clip = (int)(clip_limit * (double)(th * tw) / HIST16);
        batched = clip <= 65535;                                   // (clip = max(clip, 1) >= 1)
    }
    SHG_REQUIRE(batched || !from, SHG_E_ARG, "shg_contrast_stats_u16: a frame source needs the batched route (contrast_stats_batches)");
    if (!batched) {
        const size_t each = workspace_bytes >= (size_t)k * per ? per : 0;       // every disk its own area if there is room, else one after the other in the same
        for (int64_t i = 0; i < k; ++i)
            if (int e = shg_contrast_stats_u16(host_frames[i], h, w, pitch, clip_limit, tiles, host_cl1[i], cl1_pitch, ranks_frame2, ranks_cl13,
                                               out5 + 5 * i, static_cast<char*>(workspace) + (size_t)i * each, each ? each : workspace_bytes, stream))
                return e;
        return 0;
    }
    hipStream_t st = shg::as_stream(stream);
    const size_t c = (shg_clahe_workspace_bytes_for(h, w, tiles, 2) + 255) / 256 * 256, s2 = (shg_select_u16_workspace_bytes(2) + 255) / 256 * 256;
    Ranks8 ranks = {};
    for (int i = 0; i < 2; ++i) {
        SHG_REQUIRE(ranks_frame2[i] >= 0 && ranks_frame2[i] < h * w, SHG_E_ARG, "shg_contrast_stats_u16: rank %lld outside the image", (long long)ranks_frame2[i]);
        ranks.v[i] = ranks_frame2[i];
    }
    for (int64_t i0 = 0; i0 < k; i0 += shg::kMaxBatch) {
        const int m = (int)std::min<int64_t>(shg::kMaxBatch, k - i0);
        char* ws = static_cast<char*>(workspace) + (size_t)i0 * per;
        Disks d;
        d.src = {};
        d.dst = {};
        d.n = m;
        d.zs = per;
        for (int i = 0; i < m; ++i) {
            SHG_REQUIRE(host_frames[i0 + i] && host_cl1[i0 + i], SHG_E_ARG, "shg_contrast_stats_u16: null image");
            d.src.p[i] = host_frames[i0 + i];
            d.dst.p[i] = host_cl1[i0 + i];
            if (from) d.from.raw.p[i] = from->host_raw[i0 + i];
        }
        if (from) {
            d.fused = true;
            d.from.raw_pitch = from->raw_pitch;
            d.from.c = from->factors ? from->factors + i0 * h : nullptr;
            d.from.sx0 = from->sx0;
            d.from.dx0 = from->dx0;
            d.from.ncopy = from->ncopy;
        }
        // the selects on the CLAHE images: their slot histograms zeroed up front, every disk's in its own area
        uint32_t* sel3 = reinterpret_cast<uint32_t*>(ws + c + s2);
        const uint32_t* chunk_tile = nullptr;
        bool pass0_done = false, sel_zeroed = false;
        const RanksJob job{{ranks.v[0], ranks.v[1]}, out5 + 5 * i0, 5};
        bool ranks_done = false;
        bool window = select_window_wanted(m);
        if (int e = clahe_impl(host_frames[i0], h, w, pitch, 2, clip_limit, tiles, host_cl1[i0], cl1_pitch, ws, c, stream, &chunk_tile, sel3, (1 + 3) * 256,
                               &pass0_done, &d, &sel_zeroed, &job, &ranks_done, &window))
            return e;
        SHG_REQUIRE(chunk_tile && sel_zeroed, SHG_E_RUNTIME, "shg_contrast_stats_u16: the batched path did not take the slice histograms");
        if (!ranks_done) {
            SHG_PROF("hist_ranks", st);
            if (int e = shg::launch(k_hist_ranks, dim3(2u, 1u, (unsigned)m), dim3(1024), 0, st,
                                   HistRanksArgs{reinterpret_cast<const uint32_t*>(ws), chunk_tile, tiles * tiles, tiles * tiles, ranks, out5 + 5 * i0, per, 5}, "k_hist_ranks"))
                return e;
        }
        Disks dc = d;
        dc.src = d.dst;                                            // the selects read the CLAHE images
        if (int e = select_u16_impl(host_cl1[i0], h, w, cl1_pitch, ranks_cl13, 3, out5 + 5 * i0 + 2, sel3, shg_select_u16_workspace_bytes(3), stream, true,
                                    pass0_done, &dc, 5, window && pass0_done))
            return e;
    }
    return 0;
}
