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
