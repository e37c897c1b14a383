// SER decode: pinned host -> HBM upload of whole frames into a stack whose frame pitch may be padded
// (video_reader.py:94-123 reads the frames; here they go to the GPU once, asynchronously).
// A frame pitch that is a multiple of 8 KiB lets the frame-walking kernel read ~4 % faster than the raw
// 800 000-byte pitch of a 2000x200 16-bit frame.
#include "shg_common.h"

extern "C" int64_t shg_frame_pitch_bytes(int64_t frame_bytes) {
    if (frame_bytes <= 0) return 0;
    const int64_t unit = 8192;
    return (frame_bytes + unit - 1) / unit * unit;
}

extern "C" int shg_upload_frames(void* dst, int64_t dst_pitch_bytes, const void* host_src, int64_t frame_bytes, int64_t n_frames,
                                 shg_stream_t stream) {
    SHG_REQUIRE(dst && host_src, SHG_E_ARG, "shg_upload_frames: null pointer");
    SHG_REQUIRE(frame_bytes > 0 && n_frames > 0 && dst_pitch_bytes >= frame_bytes, SHG_E_ARG, "shg_upload_frames: bad sizes");
    hipError_t e = hipMemcpy2DAsync(dst, (size_t)dst_pitch_bytes, host_src, (size_t)frame_bytes, (size_t)frame_bytes, (size_t)n_frames,
                                    hipMemcpyHostToDevice, shg::as_stream(stream));
    if (e != hipSuccess) {
        shg::set_error("shg_upload_frames: %s", hipGetErrorString(e));
        return (int)e;
    }
    return 0;
}

// ---- uncompressed AVI frames -> the [N][Height][Width] uint8 stack --------------------------------------------
// The reference decodes AVI with cv2.VideoCapture + COLOR_BGR2GRAY (video_reader.py:68-80, 111-113).  For the
// uncompressed streams a monochrome camera produces this is a re-layout: rows bottom-up -> top-down, 4-byte row
// padding dropped, palette index -> grey, or B,G,R -> grey with OpenCV 4's 8-bit fixed-point weights.
// One pass, 1 (or 3) bytes in and 1 byte out per pixel; the raw chunks were uploaded as they lie in the file.
namespace {

__global__ __launch_bounds__(256) void k_unpack_dib(const uint8_t* __restrict__ raw, int64_t raw_pitch, int64_t height, int64_t width,
                                                    int bits, int64_t row_bytes, int bottom_up, const uint8_t* __restrict__ gray_lut,
                                                    uint8_t* __restrict__ stack, int64_t frame_stride) {
    const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;        // pixel of the frame, row-major (a frame row may be far narrower than a workgroup)
    const int64_t k = blockIdx.y;
    if (p >= height * width) return;
    const int64_t y = p / width, x = p - y * width;
    const uint8_t* row = raw + k * raw_pitch + (bottom_up ? height - 1 - y : y) * row_bytes;
    uint32_t v;
    if (bits == 24) {
        const uint32_t b = row[3 * x], g = row[3 * x + 1], r = row[3 * x + 2];
        v = (b * 3735u + g * 19235u + r * 9798u + (1u << 14)) >> 15;
    } else {
        v = row[x];
        if (gray_lut) v = gray_lut[v];
    }
    stack[k * frame_stride + y * width + x] = (uint8_t)v;
}

}  // namespace

extern "C" int shg_unpack_dib_frames(const uint8_t* raw, int64_t n_frames, int64_t raw_pitch_bytes, int64_t height, int64_t width,
                                     int bits, int64_t row_bytes, int bottom_up, const uint8_t* gray_lut, uint8_t* stack,
                                     int64_t frame_stride_px, shg_stream_t stream) {
    SHG_REQUIRE(raw && stack, SHG_E_ARG, "shg_unpack_dib_frames: null pointer");
    SHG_REQUIRE(n_frames > 0 && height > 0 && width > 0, SHG_E_ARG, "shg_unpack_dib_frames: empty input");
    SHG_REQUIRE(bits == 8 || bits == 24, SHG_E_UNSUPPORTED, "shg_unpack_dib_frames: %d-bit frames (8 or 24 only)", bits);
    SHG_REQUIRE(row_bytes >= width * (bits / 8) && raw_pitch_bytes >= row_bytes * height, SHG_E_ARG, "shg_unpack_dib_frames: bad pitches");
    SHG_REQUIRE(frame_stride_px == 0 || frame_stride_px >= height * width, SHG_E_ARG, "shg_unpack_dib_frames: frame stride smaller than a frame");
    SHG_REQUIRE(n_frames < 65536 && height * width < (1ll << 39), SHG_E_UNSUPPORTED, "shg_unpack_dib_frames: more than 65535 frames per call");
    const int64_t fstride = frame_stride_px > 0 ? frame_stride_px : height * width;
    dim3 grid((unsigned)((height * width + 255) / 256), (unsigned)n_frames);
    hipStream_t st = shg::as_stream(stream);
    SHG_PROF("unpack_dib", st);
    k_unpack_dib<<<grid, 256, 0, st>>>(raw, raw_pitch_bytes, height, width, bits, row_bytes, bottom_up, gray_lut, stack, fstride);
    return shg::check_launch("k_unpack_dib");
}
