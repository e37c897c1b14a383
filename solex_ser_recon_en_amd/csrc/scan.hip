// One scan, one call: the loop body of solex_do_work (Solex_recon.py:33-42) below the caller's interpreter.
//
//   solex_read            Solex_recon.py:50-83     shg_stage_mean_fit, shg_stage_extract
//   solex_process         Solex_recon.py:94-134    the limb fit of the first disk (shg_stage_limb_fit) or the fixed ratio /
//                                                  slant of the options, the warp of every requested disk
//   single_image_process  Solex_recon.py:136-174   the crop plan, shg_stage_process_frames
//
// The stage composites (stages.hip) already keep the interpreter out from between the kernels of ONE stage; what was left
// between the stages -- shift bookkeeping, output allocation, a dozen small conversions -- still cost a scan 0.2 ms of
// interpreter lock, which four scan workers queue for.  Here that glue is C++ as well: a scan worker makes one call per
// file and holds no lock while it runs.  Nothing is computed differently: the same composites, in the same order.
#include <math.h>
#include <string.h>
#include <atomic>
#include <vector>
#include "shg_common.h"

namespace {

constexpr size_t kAlign = 256;
inline size_t up(size_t b) { return (b + kAlign - 1) / kAlign * kAlign; }
inline int64_t round64(int64_t v) { return (v + 63) / 64 * 64; }
inline int64_t floor_div(int64_t a, int64_t b) { int64_t q = a / b; return (a % b != 0 && ((a < 0) != (b < 0))) ? q - 1 : q; }

// math.degrees / math.radians (CPython mathmodule.c: x * (180 / pi), x * (pi / 180))
const double kRadToDeg = 180.0 / 3.141592653589793238462643383279502884;
const double kDegToRad = 3.141592653589793238462643383279502884 / 180.0;

std::atomic<shg_savgol_taps_fn> g_taps_fn{nullptr};

#define SCAN_TRY(expr)                  \
    do {                                \
        if (int e_ = (expr)) return e_; \
    } while (0)

struct Regions {                       // byte sizes of the sub-areas of the workspace / the staging area, in this order
    size_t mean_ws, extract_ws, limb_ws, mean_pin, extract_pin, limb_pin, process_pin;
};

inline int64_t slit_rows(const shg_scan_request* rq) { return rq->width > rq->height ? rq->width : rq->height; }
inline int64_t spectral_cols(const shg_scan_request* rq) { return rq->width > rq->height ? rq->height : rq->width; }

inline int64_t n_requested(const shg_scan_request* rq) {
    int64_t k = 0;
    for (int i = 0; i < rq->n_shifts; ++i) k += rq->host_requested[i] ? 1 : 0;
    return k;
}

inline bool fits_limb(const shg_scan_request* rq) { return isnan(rq->ratio_fixe) && isnan(rq->slant_fix_deg); }

Regions regions(const shg_scan_request* rq) {
    Regions r;
    const int64_t ih = slit_rows(rq);
    r.mean_ws = up(shg_stage_mean_fit_workspace_bytes(rq->n_frames, rq->height, rq->width, rq->bytes_per_px));
    r.mean_pin = up(shg_stage_mean_fit_host_bytes(rq->height, rq->width));
    r.extract_ws = r.extract_pin = up(shg_stage_extract_workspace_bytes(rq->height, rq->width, rq->n_shifts));
    const bool limb = fits_limb(rq);
    r.limb_ws = limb ? up(shg_stage_limb_points_workspace_bytes(ih, rq->n_frames)) : 0;
    r.limb_pin = limb ? up(shg_stage_limb_points_host_bytes(ih, rq->n_frames)) : 0;
    r.process_pin = up(shg_stage_process_host_bytes(n_requested(rq), ih));
    return r;
}

bool no_circle(const double* c) { return c[0] == -1 && c[1] == -1 && c[2] == -1; }           // cercle == (-1, -1, -1)

// What single_image_process derives from the geometry before any pixel moves (Solex_recon.py:142-171, solex_util.py:
// 400, 542-547): the transversalium window and its taps, the crop plan, the circle after the crop, the disc.
struct Plan {
    double trans_circle[3], trans_borders[4];
    std::vector<double> taps;
    const double* taps_ptr = nullptr;
    int64_t window = 0;
    int64_t crop_w = 0, sx0 = 0, dx0 = 0, ncopy = 0;
    int64_t disc[3] = {0, 0, 0};
};

int make_plan(const shg_scan_request* rq, shg_scan_result* rs, Plan* pl) {
    const int64_t h = rs->out_h, w = rs->out_w;
    const double* c0 = rs->circle3;
    if (rq->transversalium) {
        if (!no_circle(c0)) {
            memcpy(pl->trans_circle, c0, sizeof(pl->trans_circle));
            memcpy(pl->trans_borders, rs->borders4, sizeof(pl->trans_borders));
        } else {                                              // no limb fit: the sunlit rows the line fit found (:146)
            const double c[3] = {0, 0, 99999}, b[4] = {0, (double)(rs->y1 + 20), (double)(w - 1), (double)(rs->y2 - 20)};
            memcpy(pl->trans_circle, c, sizeof(c));
            memcpy(pl->trans_borders, b, sizeof(b));
        }
        const double* c = pl->trans_circle;
        const double lo = ceil(fmax(c[1] - c[2], pl->trans_borders[1])), hi = floor(fmin(c[1] + c[2], pl->trans_borders[3]));
        SHG_REQUIRE(fabs(lo) < 9e15 && fabs(hi) < 9e15, SHG_E_VALUE, "cannot convert float NaN / infinity to integer (transversalium rows)");
        const int64_t y1 = (int64_t)lo, y2 = (int64_t)hi;
        const int64_t n = y2 - y1 > 1 ? y2 - y1 : 1;
        const int64_t odd = floor_div(n, 2) * 2 - 1;          // solex_util.py:400
        pl->window = rq->trans_strength < odd ? rq->trans_strength : odd;
        if (rq->host_taps && pl->window == rq->taps_window) {
            pl->taps_ptr = rq->host_taps;
        } else {
            shg_savgol_taps_fn fn = g_taps_fn.load();
            SHG_REQUIRE(fn, SHG_E_UNSUPPORTED, "shg_scan_file: Savitzky-Golay window %lld but the request carries taps for %lld and no "
                        "shg_host_set_savgol_taps callback is registered", (long long)pl->window, (long long)rq->taps_window);
            pl->taps.assign((size_t)(pl->window > 0 ? pl->window : 1), 0.0);
            if (int e = fn(pl->window, pl->taps.data())) {
                shg::set_error("savgol_coeffs(%lld, 3) failed", (long long)pl->window);
                return e;
            }
            pl->taps_ptr = pl->taps.data();
        }
    }
    rs->window = pl->window;
    // ---- the crop / pad block (Solex_recon.py:155-171) as numbers: new[:, dx0:dx0+n] = img[:, lo:lo+n], fill img[0, 0]
    memcpy(rs->circle_out3, c0, sizeof(rs->circle_out3));
    if (rq->has_fixed_width || rq->crop_square) {
        const int64_t nw = rq->has_fixed_width ? rq->fixed_width : h;
        SHG_REQUIRE(nw > 0, SHG_E_VALUE, "negative dimensions are not allowed (crop width %lld)", (long long)nw);
        const int64_t nw2 = floor_div(nw, 2);
        const int64_t cx = no_circle(c0) ? floor_div(w, 2) : (int64_t)c0[0];
        const int64_t tx = nw2 - cx;
        const int64_t lo = cx - nw2 > 0 ? cx - nw2 : 0, hi = cx + nw2 < w ? cx + nw2 : w;
        SHG_REQUIRE(hi >= lo, SHG_E_VALUE, "crop window [%lld, %lld) lies outside the %lld px wide image", (long long)(cx - nw2),
                    (long long)(cx + nw2), (long long)w);
        int64_t n = hi - lo;
        const int64_t dx0 = tx > 0 ? tx : 0;
        if (dx0 + n > nw) n = nw - dx0;                       // np.roll would wrap these columns round and the refill overwrite them
        pl->crop_w = nw; pl->sx0 = lo; pl->dx0 = dx0; pl->ncopy = n;
        if (!no_circle(c0)) rs->circle_out3[0] = (double)nw2;
    }
    rs->crop4[0] = pl->crop_w; rs->crop4[1] = pl->sx0; rs->crop4[2] = pl->dx0; rs->crop4[3] = pl->ncopy;
    // ---- cv2.circle(frame_protus, (x0, y0), r, 80, -1) (solex_util.py:542-547)
    const double* c = rs->circle_out3;
    if (!no_circle(c) && rq->disk_display) {
        SHG_REQUIRE(fabs(c[0]) < 9e15 && fabs(c[1]) < 9e15 && fabs(c[2]) < 9e15, SHG_E_VALUE, "cannot convert float NaN / infinity to integer (disk circle)");
        const int64_t r = (int64_t)c[2] + rq->delta_radius;
        if (r > 0) { pl->disc[0] = (int64_t)c[0]; pl->disc[1] = (int64_t)c[1]; pl->disc[2] = r; }
    }
    memcpy(rs->disc3, pl->disc, sizeof(rs->disc3));
    return 0;
}

}  // namespace

extern "C" int shg_host_set_savgol_taps(shg_savgol_taps_fn fn) {
    g_taps_fn.store(fn);
    return 0;
}

extern "C" size_t shg_scan_workspace_bytes(const shg_scan_request* rq) {
    if (!rq || rq->height <= 0 || rq->width <= 0 || rq->n_frames <= 0 || rq->n_shifts <= 0 || !rq->host_requested) return 0;
    const Regions r = regions(rq);
    return r.mean_ws + r.extract_ws + r.limb_ws + kAlign;
}

extern "C" size_t shg_scan_host_bytes(const shg_scan_request* rq) {
    if (!rq || rq->height <= 0 || rq->width <= 0 || rq->n_frames <= 0 || rq->n_shifts <= 0 || !rq->host_requested) return 0;
    const Regions r = regions(rq);
    return r.mean_pin + r.extract_pin + r.limb_pin + r.process_pin + kAlign;
}

// Pass A of a scan that will run later (the scan pool calls this when the scan is submitted): launched on the device's frame-pass
// lane now, found there by the scan's first stage.  Nothing happens -- and nothing is wrong -- without a lane, for a resumed scan,
// or for a request shg_scan_file is going to refuse.
extern "C" int shg_scan_prelaunch(const shg_scan_request* rq, shg_stream_t after, int* launched) {
    SHG_REQUIRE(rq && launched, SHG_E_ARG, "shg_scan_prelaunch: null pointer");
    *launched = 0;
    if (rq->struct_bytes != sizeof(shg_scan_request) || rq->start_phase != 0 || !rq->stack || !rq->workspace || rq->n_frames <= 0 ||
        rq->height <= 0 || rq->width <= 0 || rq->n_shifts <= 0 || (rq->bytes_per_px != 1 && rq->bytes_per_px != 2))
        return 0;
    const Regions rg = regions(rq);
    const size_t acc = shg_accumulate_workspace_bytes(rq->n_frames, rq->height, rq->width, rq->bytes_per_px);
    if (rq->workspace_bytes < rg.mean_ws || acc == 0 || acc > rg.mean_ws) return 0;
    // the partials are the first thing shg_stage_mean_fit takes from its workspace, which is the first region of the scan's
    return shg_pass_a_prelaunch(rq->stack, rq->n_frames, rq->height, rq->width, rq->bytes_per_px, rq->frame_stride_px, rq->workspace, acc, after,
                                launched);
}

extern "C" int shg_scan_file(const shg_scan_request* rq, shg_scan_result* rs, shg_stream_t stream) {
    SHG_HOST_TIME("scan_file");
    SHG_REQUIRE(rq && rs, SHG_E_ARG, "shg_scan_file: null pointer");
    SHG_REQUIRE(rq->struct_bytes == sizeof(shg_scan_request), SHG_E_ARG, "shg_scan_file: the caller's shg_scan_request has %u bytes, the library's %zu",
                rq->struct_bytes, sizeof(shg_scan_request));
    SHG_REQUIRE(rq->stack && rq->host_shifts && rq->host_requested && rq->mean_out && rq->max_out && rq->disks && rq->minmax_slots &&
                rq->workspace && rq->host_pinned && rq->host_fit, SHG_E_ARG, "shg_scan_file: null pointer in the request");
    SHG_REQUIRE(rq->n_frames > 0 && rq->height > 0 && rq->width > 0 && rq->n_shifts > 0, SHG_E_ARG, "shg_scan_file: empty input");
    SHG_REQUIRE(rq->disk_pitch >= rq->n_frames && rq->disk_plane_stride > 0, SHG_E_ARG, "shg_scan_file: bad disk layout");
    SHG_REQUIRE(rq->start_phase == 0 || rq->start_phase == 3, SHG_E_ARG, "shg_scan_file: start_phase must be 0 or 3");
    const int64_t ih = slit_rows(rq), n_cols = rq->n_frames;
    const int S = rq->n_shifts;
    const int64_t k = n_requested(rq);
    const bool limb = fits_limb(rq);
    SHG_REQUIRE(!limb || (rq->host_gauss_taps && rq->host_points && rq->host_flags && rq->points_cap > 0), SHG_E_ARG,
                "shg_scan_file: the limb fit needs host_gauss_taps, host_points, host_flags");
    const Regions rg = regions(rq);
    char* ws = static_cast<char*>(rq->workspace);
    char* pin = static_cast<char*>(rq->host_pinned);
    const size_t ws_fixed = rg.mean_ws + rg.extract_ws + rg.limb_ws;
    SHG_REQUIRE(rq->workspace_bytes >= ws_fixed, SHG_E_WORKSPACE, "shg_scan_file: workspace %zu < %zu bytes", rq->workspace_bytes, ws_fixed);
    SHG_REQUIRE(rq->host_pinned_bytes >= rg.mean_pin + rg.extract_pin + rg.limb_pin + rg.process_pin, SHG_E_WORKSPACE,
                "shg_scan_file: pinned staging area %zu < %zu bytes", rq->host_pinned_bytes, rg.mean_pin + rg.extract_pin + rg.limb_pin + rg.process_pin);
    char* mean_ws = ws;
    char* extract_ws = mean_ws + rg.mean_ws;
    char* limb_ws = extract_ws + rg.extract_ws;
    char* process_ws = limb_ws + rg.limb_ws;
    char* mean_pin = pin;
    char* extract_pin = mean_pin + rg.mean_pin;
    char* limb_pin = extract_pin + rg.extract_pin;
    char* process_pin = limb_pin + rg.limb_pin;

    if (rq->start_phase == 0) {
        memset(rs, 0, sizeof(*rs));
        rs->fit_image_off = rs->frames_off = rs->detrans_off = rs->products_off = rs->results_off = -1;
        // ---- solex_read (Solex_recon.py:50-83) ----
        int64_t y12[2];
        SCAN_TRY(shg_stage_mean_fit(rq->stack, rq->n_frames, rq->height, rq->width, rq->bytes_per_px, rq->frame_stride_px, nullptr, nullptr,
                                    rq->n_frames, rq->mean_out, rq->max_out, y12, rs->p4, rq->host_fit, rq->host_trace_sharp, rq->host_mask_good,
                                    mean_ws, rg.mean_ws, mean_pin, rg.mean_pin, stream));
        rs->y1 = y12[0];
        rs->y2 = y12[1];
        rs->phase_done = 1;
        if (limb) {                                            // the limb stage's accumulators are cleared by the extraction's last launch
            shg::t_zero_with_fold = limb_ws;
            shg::t_zero_with_fold_words = shg::limb_prepare_zero_words(ih, n_cols);
        }
        const int extract_status = shg_stage_extract(rq->stack, rq->n_frames, rq->height, rq->width, rq->bytes_per_px, rq->frame_stride_px, rq->host_fit,
                                                     rq->host_shifts, S, rq->disks, rq->disk_pitch, rq->disk_plane_stride, n_cols, 0, rq->flip_x,
                                                     rq->minmax_slots, extract_ws, rg.extract_ws, extract_pin, rg.extract_pin, stream);
        shg::t_zero_with_fold = nullptr;                       // (whatever happened: the request does not outlive this call)
        shg::t_zero_with_fold_words = 0;
        if (extract_status) { shg::t_prezeroed = nullptr; return extract_status; }
        rs->phase_done = 2;
        // ---- solex_process: the geometry (Solex_recon.py:104-122) ----
        bool later_disks = false;                              // a requested disk other than the first goes through correct_image with
        for (int i = 1; i < S; ++i) later_disks = later_disks || rq->host_requested[i];     // the angle options['slant_fix'] holds
        double mat3[9], inv[4], origin[2], det;
        if (limb) {
            int64_t dims[2];
            const int limb_status = shg_stage_limb_fit(rq->disks, ih, n_cols, rq->disk_pitch, rq->host_gauss_taps, rq->host_points, rq->host_flags, rq->points_cap,
                                                       rs->counts3, rs->geom16, dims, rq->host_outline200, limb_ws, rg.limb_ws, limb_pin, rg.limb_pin, stream);
            shg::t_prezeroed = nullptr;                        // (spent by shg_limb_prepare, or never looked at: the other limb kernels)
            if (limb_status) return limb_status;
            const double* g = rs->geom16;
            rs->limb_fitted = 1;
            rs->phi = g[3];
            rs->ratio = g[4];
            memcpy(rs->circle3, g + 5, 3 * sizeof(double));
            memcpy(rs->borders4, g + 8, 4 * sizeof(double));
            memcpy(rs->h_first, g + 12, 3 * sizeof(double));
            rs->theta_first = g[15];
            rs->out_h = dims[0];
            rs->out_w = dims[1];
            memcpy(rs->h_rest, rs->h_first, sizeof(rs->h_rest));
            rs->theta_rest = rs->theta_first;
            if (later_disks) {
                const double phi_again = (rs->phi * kRadToDeg) * kDegToRad;      // math.radians(options['slant_fix']), :117-121
                int64_t oh, ow;
                SCAN_TRY(shg_host_warp_geometry(phi_again, rs->ratio, ih, n_cols, mat3, inv, origin, &det, &rs->theta_rest, &oh, &ow));
                SHG_REQUIRE(oh == rs->out_h && ow == rs->out_w, SHG_E_VALUE, "process_frames: the frames must share one shape and row pitch");
                memcpy(rs->h_rest, mat3, 3 * sizeof(double));
            }
        } else {
            rs->ratio = isnan(rq->ratio_fixe) ? 1.0 : rq->ratio_fixe;
            rs->phi = isnan(rq->slant_fix_deg) ? 0.0 : rq->slant_fix_deg * kDegToRad;
            SCAN_TRY(shg_host_warp_geometry(rs->phi, rs->ratio, ih, n_cols, mat3, inv, origin, &det, &rs->theta_first, &rs->out_h, &rs->out_w));
            memcpy(rs->h_first, mat3, 3 * sizeof(double));
            memcpy(rs->h_rest, mat3, 3 * sizeof(double));
            rs->theta_rest = rs->theta_first;
            rs->circle3[0] = rs->circle3[1] = rs->circle3[2] = -1;
            shg::t_prezeroed = nullptr;
        }
        rs->phase_done = 3;
    } else {
        SHG_REQUIRE(rs->phase_done == 3, SHG_E_ARG, "shg_scan_file: nothing to resume (phase_done = %d)", rs->phase_done);
    }

    // ---- single_image_process: what follows from the geometry (Solex_recon.py:136-171) ----
    Plan pl;
    SCAN_TRY(make_plan(rq, rs, &pl));
    const int64_t out_h = rs->out_h, out_w = rs->out_w;
    SHG_REQUIRE(out_h > 0 && out_w > 0, SHG_E_VALUE, "correct_image: empty output (%lld x %lld)", (long long)out_h, (long long)out_w);
    const int64_t frame_pitch = round64(out_w);
    const int64_t prod_w = pl.crop_w > 0 ? pl.crop_w : out_w, prod_pitch = round64(prod_w);
    const bool fit_image = rq->want_fit_image && !rq->host_requested[0];
    const bool detrans = rq->transversalium && rq->keep_detrans && k > 0;
    const size_t frame_bytes = up((size_t)out_h * frame_pitch * 2), prod_bytes = up((size_t)out_h * prod_pitch * 2);
    size_t off = 0;
    rs->fit_image_off = fit_image ? (int64_t)off : -1;
    off += fit_image ? frame_bytes : 0;
    rs->frames_off = k > 0 ? (int64_t)off : -1;
    off += (size_t)k * frame_bytes;
    rs->detrans_off = detrans ? (int64_t)off : -1;
    off += detrans ? (size_t)k * frame_bytes : 0;
    rs->products_off = k > 0 ? (int64_t)off : -1;
    off += (size_t)k * 3 * prod_bytes;
    rs->results_off = k > 0 ? 0 : -1;
    rs->needed_results_bytes = (size_t)k * 2 * prod_bytes;
    rs->frame_pitch = frame_pitch;
    rs->n_out = k;
    rs->prod_w = prod_w;
    rs->prod_pitch = prod_pitch;
    rs->needed_arena_bytes = off;
    const size_t process_ws_bytes = k > 0 ? up(shg_stage_process_workspace_bytes(k, out_h, out_w, pl.crop_w, rq->tiles)) : 0;
    rs->needed_workspace_bytes = ws_fixed + process_ws_bytes;
    if (rq->arena_bytes < rs->needed_arena_bytes || (off > 0 && !rq->arena) || rq->results_bytes < rs->needed_results_bytes ||
        (k > 0 && !rq->results) || rq->workspace_bytes < rs->needed_workspace_bytes) {
        shg::set_error("shg_scan_file: the corrected images are %lld x %lld: arena %zu of %zu bytes, results %zu of %zu bytes, workspace %zu of "
                       "%zu bytes -- grow them and call again with start_phase = 3", (long long)out_h, (long long)out_w, rq->arena_bytes,
                       rs->needed_arena_bytes, rq->results_bytes, rs->needed_results_bytes, rq->workspace_bytes, rs->needed_workspace_bytes);
        return SHG_E_WORKSPACE;
    }
    SHG_REQUIRE(((reinterpret_cast<uintptr_t>(rq->arena) | reinterpret_cast<uintptr_t>(rq->results)) & 255) == 0, SHG_E_ARG,
                "shg_scan_file: arenas not 256-byte aligned");
    char* arena = static_cast<char*>(rq->arena);
    char* results = static_cast<char*>(rq->results);
    const uint32_t* extrema = rq->minmax_slots + (size_t)S * 128;                  // {min, max} per plane, after the slots
    // every warp of the file in one launch: the requested disks and, for the diagnostic plot, the ellipse-fit disk
    std::vector<const uint16_t*> w_src;
    std::vector<uint16_t*> w_dst;
    std::vector<const uint32_t*> w_mm;
    std::vector<double> w_h;
    auto add_warp = [&](int i, uint16_t* dst) {
        const double* hr = i == 0 ? rs->h_first : rs->h_rest;
        w_src.push_back(rq->disks + (size_t)i * rq->disk_plane_stride);
        w_dst.push_back(dst);
        w_mm.push_back(extrema + 2 * i);
        w_h.insert(w_h.end(), hr, hr + 3);
    };
    if (fit_image) add_warp(0, reinterpret_cast<uint16_t*>(arena + rs->fit_image_off));
    std::vector<const uint16_t*> frames((size_t)k);
    std::vector<uint16_t*> det((size_t)k), prod[5];
    for (auto& v : prod) v.resize((size_t)k);
    int64_t j = 0;
    for (int i = 0; i < S; ++i) {
        if (!rq->host_requested[i]) continue;
        uint16_t* f = reinterpret_cast<uint16_t*>(arena + rs->frames_off + (size_t)j * frame_bytes);
        add_warp(i, f);
        frames[j] = f;
        if (detrans) det[j] = reinterpret_cast<uint16_t*>(arena + rs->detrans_off + (size_t)j * frame_bytes);
        for (int p = 0; p < 3; ++p) prod[p][j] = reinterpret_cast<uint16_t*>(arena + rs->products_off + ((size_t)j * 3 + p) * prod_bytes);
        for (int p = 0; p < 2; ++p) prod[3 + p][j] = reinterpret_cast<uint16_t*>(results + ((size_t)j * 2 + p) * prod_bytes);
        ++j;
    }
    if (!w_src.empty())
        SCAN_TRY(shg::warp_rows_batch(w_src.data(), (int64_t)w_src.size(), ih, n_cols, rq->disk_pitch, w_h.data(), w_dst.data(), out_h, out_w, frame_pitch,
                                      w_mm.data(), stream));
    if (k > 0) {
        SCAN_TRY(shg_stage_process_frames(frames.data(), k, out_h, out_w, frame_pitch, rq->transversalium, pl.trans_circle, pl.trans_borders,
                                          pl.taps_ptr, pl.window, rq->host_factors, pl.crop_w, pl.sx0, pl.dx0, pl.ncopy, rq->clip_limit, rq->tiles,
                                          pl.disc[0], pl.disc[1], pl.disc[2], detrans ? det.data() : nullptr, frame_pitch, prod[0].data(),
                                          prod[1].data(), prod[2].data(), prod[3].data(), prod[4].data(), prod_pitch, process_ws, process_ws_bytes,
                                          process_pin, rg.process_pin, stream));
    }
    rs->phase_done = 4;
    return 0;
}
