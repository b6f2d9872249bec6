// api.hip -- context management and the C-ABI entry points of libwassgpu.
#include "common.h"

#include <stdarg.h>
#include <new>
#include <vector>

namespace wass {

int set_err(wass_ctx* c, int code, const char* fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (c) c->err = buf;
    return code;
}

int ensure(wass_ctx* c, Buf& b, size_t bytes)
{
    if (bytes <= b.cap) return WASS_OK;
    if (b.p) { (void)hipFree(b.p); b.p = nullptr; b.cap = 0; }
    // round up so that slightly different frame sizes do not thrash the allocator
    const size_t want = (bytes + ((size_t)1 << 20) - 1) & ~(((size_t)1 << 20) - 1);
    hipError_t e = hipMalloc(&b.p, want);
    if (e != hipSuccess) {
        b.p = nullptr;
        return set_err(c, WASS_ERR_NO_MEMORY, "hipMalloc(%zu) failed: %s", want, hipGetErrorString(e));
    }
    b.cap = want;
    return WASS_OK;
}

static void release(Buf& b)
{
    if (b.p) (void)hipFree(b.p);
    b.p = nullptr; b.cap = 0;
}

// derived parameters
static int make_dims(wass_ctx* c, int w, int h, const wass_sgm_params* p, SgmDims& d)
{
    if (!p || w <= 0 || h <= 0) return set_err(c, WASS_ERR_INVALID_ARG, "bad image size %dx%d", w, h);
    if (p->num_disp <= 0 || (p->num_disp % 16) != 0)
        return set_err(c, WASS_ERR_INVALID_ARG, "MAX_DISPARITY must be a positive multiple of 16 (got %d)", p->num_disp);
    if (p->num_disp > 1024) return set_err(c, WASS_ERR_UNSUPPORTED, "MAX_DISPARITY %d > 1024", p->num_disp);
    if (p->min_disp < 0) return set_err(c, WASS_ERR_UNSUPPORTED, "negative MIN_DISPARITY is not supported");
    if (p->ndirs != 5 && p->ndirs != 8) return set_err(c, WASS_ERR_INVALID_ARG, "ndirs must be 5 or 8");
    d.speckle_win = p->speckle_win > 0 ? p->speckle_win : 0;
    d.speckle_range = p->speckle_range;
    const int win = p->win > 0 ? p->win : 5;
    if (win > 17) return set_err(c, WASS_ERR_UNSUPPORTED, "WINSIZE %d > 17 is not supported", win);
    d.w = w; d.h = h; d.D = p->num_disp;
    d.off_pos = p->disp_offset > 0 ? p->disp_offset : 0;
    d.comp = p->disp_offset > 0 ? 0 : -p->disp_offset;
    if (d.D + d.off_pos - d.comp < 0)
        return set_err(c, WASS_ERR_INVALID_ARG, "DISPARITY_OFFSET %d exceeds MAX_DISPARITY", p->disp_offset);
    d.Wp = w + d.D + d.off_pos;
    d.NP = (d.D + 127) / 128;
    d.Dp = 128 * d.NP;
    d.minD = p->min_disp; d.maxD = d.minD + d.D;
    d.minX1 = d.maxD;
    d.width1 = d.Wp - d.maxD;
    d.SW2 = win / 2;
    d.P1 = p->P1 > 0 ? p->P1 : 2;
    d.P2 = p->P2 > 0 ? p->P2 : 5;
    if (d.P2 < d.P1 + 1) d.P2 = d.P1 + 1;
    if (d.P2 > 32767) return set_err(c, WASS_ERR_INVALID_ARG, "P2 %d does not fit the int16 cost type", d.P2);
    d.uniq = p->uniq_ratio >= 0 ? p->uniq_ratio : 10;
    d.d12 = p->disp12_max_diff > 0 ? p->disp12_max_diff : 1;
    d.ftzero = (p->prefilter_cap > 15 ? p->prefilter_cap : 15) | 1;
    if (d.ftzero > 127) return set_err(c, WASS_ERR_UNSUPPORTED, "DENSE_PREFILTER_CAP %d > 126", p->prefilter_cap);
    d.ndirs = p->ndirs;
    if (d.width1 <= d.SW2) return set_err(c, WASS_ERR_INVALID_ARG, "image too narrow for the matching window");
    return WASS_OK;
}

}  // namespace wass

using namespace wass;

extern "C" {

const char* wass_version(void) { return "wass_amd 0.1 (gfx950)"; }

// The HIP runtime multiplexes a process's streams onto GPU_MAX_HW_QUEUES hardware queues (default 4), and a hardware queue is in-order.
// A context has four streams (SGM, side, copy, tail) and the process has its null stream: with four queues two of them share one, and
// WHICH two is decided by the runtime among equally loaded queues -- it differs from run to run.  When the tail stream lands on the side
// stream's queue the anti-diagonal checkpoint sweep waits behind the previous frame's tail kernels (aggregation 6.2 instead of 5.85 ms);
// on the SGM stream's queue the frame period becomes SGM + tail (115 instead of 130 pairs/s); with the null or the copy stream it is
// harmless.  Round 6 found this as the "slow boxes" of four rounds (profiles/r06_x_streams.log -- made with a knob, since removed, that
// created dummy streams in front of the context's to shift the assignment -- and r06_x_hwqueues.log).  Six queues: every
// stream of a context has its own (5 and 6 measure the same, 8 is 1.5 % slower).  Only effective before the runtime initialises -- i.e.
// in the shipped executables, whose first HIP call is made here; bench.py sets the same default before it imports PyTorch.
static void default_hw_queues() { (void)setenv("GPU_MAX_HW_QUEUES", "6", 0); }

int wass_device_count(int* n_devices)
{
    if (!n_devices) return WASS_ERR_INVALID_ARG;
    default_hw_queues();
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n < 0) n = 0;
    *n_devices = n;
    return WASS_OK;
}

int wass_ctx_create(int device_id, wass_ctx** out)
{
    if (!out) return WASS_ERR_INVALID_ARG;
    *out = nullptr;
    default_hw_queues();
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device_id < 0 || device_id >= n) return WASS_ERR_DEVICE;
    if (hipSetDevice(device_id) != hipSuccess) return WASS_ERR_DEVICE;
    // Host threads that wait for the GPU sleep instead of spinning (hipDeviceScheduleAuto spins while there are fewer contexts than cores,
    // and the hipEventBlockingSync flag of an event alone did not change that): a sequence driver's submitting thread spent 5 ms of CPU per
    // config-B frame waiting for the frame before last -- 33.5 -> 27.8 ms of host CPU per frame, same frames/s (round 6,
    // scripts/host_cost.py).  Takes effect when this is the first HIP call of the process on that device (the shipped executables; in a
    // process whose runtime is already up -- bench.py under PyTorch -- the call is refused and nothing changes).  WASS_BLOCKING_SYNC=0: spin.
    {
        const char* e = getenv("WASS_BLOCKING_SYNC");
        if (!e || atoi(e) != 0) { (void)hipSetDeviceFlags(hipDeviceScheduleBlockingSync); (void)hipGetLastError(); }
    }
    wass_ctx* c = new (std::nothrow) wass_ctx();
    if (!c) return WASS_ERR_NO_MEMORY;
    c->device = device_id;
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { wass_ctx_destroy(c); return WASS_ERR_DEVICE; }
    if (hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking) != hipSuccess) { wass_ctx_destroy(c); return WASS_ERR_DEVICE; }
    if (hipStreamCreateWithFlags(&c->copy, hipStreamNonBlocking) != hipSuccess) { wass_ctx_destroy(c); return WASS_ERR_DEVICE; }
    if (hipStreamCreateWithFlags(&c->tail, hipStreamNonBlocking) != hipSuccess) { wass_ctx_destroy(c); return WASS_ERR_DEVICE; }
    if (hipEventCreateWithFlags(&c->ev_post, hipEventDisableTiming) != hipSuccess) { wass_ctx_destroy(c); return WASS_ERR_DEVICE; }
    if (hipEventCreateWithFlags(&c->ev_pack, hipEventDisableTiming) != hipSuccess) { wass_ctx_destroy(c); return WASS_ERR_DEVICE; }
    for (auto& fs : c->fslot)
        if (hipEventCreateWithFlags(&fs.ev_copy, hipEventDisableTiming | (getenv("WASS_SPIN_WAIT") ? 0 : hipEventBlockingSync)) != hipSuccess) { wass_ctx_destroy(c); return WASS_ERR_DEVICE; }
    c->ev_copy = c->fslot[0].ev_copy;                      // "the most recently recorded download event": an alias of one of the two
    for (auto& set : c->evs)
        for (auto& e : set)
            if (hipEventCreate(&e) != hipSuccess) { wass_ctx_destroy(c); return WASS_ERR_DEVICE; }
    if (hipEventCreateWithFlags(&c->ev_cost, hipEventDisableTiming) != hipSuccess) { wass_ctx_destroy(c); return WASS_ERR_DEVICE; }
    for (auto& e : c->ev_ckpt)
        if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) { wass_ctx_destroy(c); return WASS_ERR_DEVICE; }
    if (ensure(c, c->flags, 64) != WASS_OK) { wass_ctx_destroy(c); return WASS_ERR_NO_MEMORY; }
    if (hipHostMalloc((void**)&c->h_flags, 64, hipHostMallocDefault) != hipSuccess) { wass_ctx_destroy(c); return WASS_ERR_NO_MEMORY; }
    for (int k = 0; k < 16; ++k) c->h_flags[k] = 0;       // one status word per timing set, 16 bytes apart
    mesh_pool_ctx_alive(c, true);
    *out = c;
    return WASS_OK;
}

void wass_ctx_destroy(wass_ctx* c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->tail) (void)hipStreamSynchronize(c->tail);
    coll_release(c);
    mesh_pool_ctx_alive(c, false);
    mesh_pool_purge(c);
    for (Buf* b : { &c->bt1, &c->bt2, &c->hsum, &c->C, &c->S, &c->ckpt, &c->sel_d16, &c->sel_key, &c->raw,
                    &c->flags, &c->tmp_in0, &c->tmp_in1, &c->tmp_out, &c->tmp_mask, &c->fA, &c->fB, &c->fC, &c->fD, &c->fE, &c->uf, &c->rs_r, &c->rs_l, &c->raw2, &c->grid, &c->scratch, &c->counters, &c->tri_cnt, &c->inl, &c->inl2, &c->clahe_lut, &c->ccmask, &c->und_cache[0].xy, &c->und_cache[1].xy, &c->dstate, &c->rect_tab, &c->rect_mx, &c->rect_my, &c->xyzc, &c->limits, &c->jpeg_scratch, &c->jpeg_huff,
                    &c->jpeg_out, &c->jpeg_info, &c->jpeg_part })
        release(*b);
    for (auto& e : c->ev_dbg) if (e) (void)hipEventDestroy(e);
    for (auto& k : c->kev) { if (k.a) (void)hipEventDestroy(k.a); if (k.b) (void)hipEventDestroy(k.b); }
    if (c->h_dbg_info) (void)hipHostFree(c->h_dbg_info);
    for (auto& set : c->evs) for (auto& e : set) if (e) (void)hipEventDestroy(e);
    for (auto& e : c->ev_ckpt) if (e) (void)hipEventDestroy(e);
    if (c->ev_cost) (void)hipEventDestroy(c->ev_cost);
    if (c->side) { (void)hipStreamSynchronize(c->side); (void)hipStreamDestroy(c->side); }
    if (c->copy) { (void)hipStreamSynchronize(c->copy); (void)hipStreamDestroy(c->copy); }
    if (c->tail) { (void)hipStreamSynchronize(c->tail); (void)hipStreamDestroy(c->tail); }
    if (c->ev_post) (void)hipEventDestroy(c->ev_post);
    if (c->h_flags) (void)hipHostFree(c->h_flags);
    for (auto& fs : c->fslot) { if (fs.h_frame) (void)hipHostFree(fs.h_frame); if (fs.ev_copy) (void)hipEventDestroy(fs.ev_copy); }
    if (c->ev_stage2) (void)hipEventDestroy(c->ev_stage2);
    if (c->h_stage) (void)hipHostFree(c->h_stage);
    if (c->ev_stage) (void)hipEventDestroy(c->ev_stage);
    if (c->ev_producer) (void)hipEventDestroy(c->ev_producer);
    if (c->ev_dl) (void)hipEventDestroy(c->ev_dl);
    if (c->ev_dl_tail) (void)hipEventDestroy(c->ev_dl_tail);
    for (auto& set : c->ev_tail_sets) for (auto& e : set) if (e) (void)hipEventDestroy(e);
    if (c->ev_pack) (void)hipEventDestroy(c->ev_pack);
    for (auto& u : c->uploads) if (u.ev) (void)hipEventDestroy(u.ev);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

const char* wass_last_error(const wass_ctx* c) { return c ? c->err.c_str() : "null context"; }
void* wass_ctx_stream(wass_ctx* c) { return c ? (void*)c->stream : nullptr; }

int wass_ctx_set_debug(wass_ctx* c, int on)
{
    if (!c) return WASS_ERR_INVALID_ARG;
    c->debug = on != 0;
    return WASS_OK;
}

int wass_ctx_synchronize(wass_ctx* c)
{
    if (!c) return WASS_ERR_INVALID_ARG;
    WASS_HIP(c, hipStreamSynchronize(c->stream));
    WASS_HIP(c, hipStreamSynchronize(c->tail));
    WASS_HIP(c, hipStreamSynchronize(c->copy));
    return WASS_OK;
}

int wass_ctx_wait_for_stream(wass_ctx* c, void* producer_stream)
{
    if (!c) return WASS_ERR_INVALID_ARG;
    WASS_HIP(c, hipSetDevice(c->device));
    // An idle producer has nothing to wait for.  This is not only a shortcut: the runtime multiplexes every stream of the
    // process onto four hardware queues, and an event recorded on an idle stream that happens to share its queue with the
    // context's tail stream completes only behind the previous frame's tail -- the SGM stage of the next frame, which
    // waits for that event, then starts after the tail instead of beside it (measured: frame period = SGM + tail).
    if (hipStreamQuery((hipStream_t)producer_stream) == hipSuccess) return WASS_OK;
    if (!c->ev_producer) WASS_HIP(c, hipEventCreateWithFlags(&c->ev_producer, hipEventDisableTiming));
    WASS_HIP(c, hipEventRecord(c->ev_producer, (hipStream_t)producer_stream));
    WASS_HIP(c, hipStreamWaitEvent(c->stream, c->ev_producer, 0));
    WASS_HIP(c, hipStreamWaitEvent(c->tail, c->ev_producer, 0));
    return WASS_OK;
}

int wass_device_alloc(wass_ctx* c, size_t nbytes, void** d_out)
{
    if (!c || !d_out || nbytes == 0) return set_err(c, WASS_ERR_INVALID_ARG, "bad argument");
    WASS_HIP(c, hipSetDevice(c->device));
    *d_out = nullptr;
    if (hipMalloc(d_out, nbytes) != hipSuccess) { *d_out = nullptr; return set_err(c, WASS_ERR_NO_MEMORY, "hipMalloc(%zu) failed", nbytes); }
    // hipMemset on device memory returns before the fill has run, and the fill is ordered on the NULL stream, which the
    // context's non-blocking streams do not wait for: without the synchronisation it can land on top of what the caller's
    // first kernel or upload wrote (seen: the first rectified crop of a sequence partly zeroed, differently every run)
    WASS_HIP(c, hipMemset(*d_out, 0, nbytes));
    WASS_HIP(c, hipDeviceSynchronize());
    return WASS_OK;
}
void wass_device_free(wass_ctx* c, void* d_ptr)
{
    if (!c || !d_ptr) return;
    (void)hipSetDevice(c->device);
    (void)hipFree(d_ptr);
}
int wass_download(wass_ctx* c, void* h_dst, const void* d_src, size_t nbytes)
{
    if (!c || !h_dst || !d_src) return set_err(c, WASS_ERR_INVALID_ARG, "null argument");
    WASS_HIP(c, hipSetDevice(c->device));
    WASS_HIP(c, hipStreamSynchronize(c->stream));
    WASS_HIP(c, hipStreamSynchronize(c->tail));
    WASS_HIP(c, hipStreamSynchronize(c->copy));
    WASS_HIP(c, hipMemcpy(h_dst, d_src, nbytes, hipMemcpyDeviceToHost));
    return WASS_OK;
}
int wass_download_async(wass_ctx* c, void* h_dst, const void* d_src, size_t nbytes)
{
    if (!c || !h_dst || !d_src) return set_err(c, WASS_ERR_INVALID_ARG, "null argument");
    WASS_HIP(c, hipSetDevice(c->device));
    if (!c->ev_dl) WASS_HIP(c, hipEventCreateWithFlags(&c->ev_dl, hipEventDisableTiming));
    WASS_HIP(c, hipEventRecord(c->ev_dl, c->stream));                 // after what the SGM stream has been given so far
    WASS_HIP(c, hipStreamWaitEvent(c->copy, c->ev_dl, 0));
    if (c->tail_overlap && c->tail) {                                 // ... and the tail stream: the source may be a map it wrote
        if (!c->ev_dl_tail) WASS_HIP(c, hipEventCreateWithFlags(&c->ev_dl_tail, hipEventDisableTiming));
        if (hipStreamQuery(c->tail) != hipSuccess) {                  // an idle stream has nothing to order (and an event recorded on
            WASS_HIP(c, hipEventRecord(c->ev_dl_tail, c->tail));      // one can land behind another stream's work: DESIGN.md 4.3)
            WASS_HIP(c, hipStreamWaitEvent(c->copy, c->ev_dl_tail, 0));
        }
    }
    WASS_HIP(c, hipMemcpyAsync(h_dst, d_src, nbytes, hipMemcpyDeviceToHost, c->copy));
    return WASS_OK;
}
int wass_pinned_alloc(wass_ctx* c, size_t nbytes, void** h_out)
{
    if (!c || !h_out || nbytes == 0) return set_err(c, WASS_ERR_INVALID_ARG, "bad argument");
    WASS_HIP(c, hipSetDevice(c->device));
    *h_out = nullptr;
    if (hipHostMalloc(h_out, nbytes, hipHostMallocDefault) != hipSuccess) { *h_out = nullptr; return set_err(c, WASS_ERR_NO_MEMORY, "hipHostMalloc(%zu) failed", nbytes); }
    return WASS_OK;
}
void wass_pinned_free(wass_ctx* c, void* h_ptr)
{
    if (!c || !h_ptr) return;
    (void)hipSetDevice(c->device);
    (void)hipHostFree(h_ptr);
}

int wass_ctx_set_tail_overlap(wass_ctx* c, int on)
{
    if (!c) return WASS_ERR_INVALID_ARG;
    WASS_HIP(c, hipStreamSynchronize(c->stream));          // switching streams mid-flight would drop the ordering
    WASS_HIP(c, hipStreamSynchronize(c->tail));
    c->tail_overlap = on != 0;
    return WASS_OK;
}

int wass_sgm_disparity_dev(wass_ctx* c, const uint8_t* d_right, const uint8_t* d_left, int w, int h, size_t pitch,
                           const wass_sgm_params* p, int16_t* d_out)
{
    if (!c || !d_right || !d_left || !d_out) return set_err(c, WASS_ERR_INVALID_ARG, "null argument");
    if (pitch < (size_t)w) return set_err(c, WASS_ERR_INVALID_ARG, "pitch %zu < width %d", pitch, w);
    if (!p || !(p->dense_scale > 0)) return set_err(c, WASS_ERR_INVALID_ARG, "DENSE_SCALE must be positive");
    WASS_HIP(c, hipSetDevice(c->device));
    int rc;
    if ((rc = wait_uploads(c, d_right, c->stream)) || (rc = wait_uploads(c, d_left, c->stream))) return rc;
    if (p->dense_scale != 1.0) {
        // wass_stereo.cpp:788-796: both crops resized with cv::resize INTER_CUBIC (x only when the scale is > 1) before the
        // padding; the disparity map comes out at that size (wass_dense_input_size)
        int ws = 0, hs = 0;
        if (wass_dense_input_size(w, h, p->dense_scale, &ws, &hs) != WASS_OK) return set_err(c, WASS_ERR_INVALID_ARG, "DENSE_SCALE %g leaves no image", p->dense_scale);
        if ((rc = ensure(c, c->rs_r, (size_t)ws * hs)) || (rc = ensure(c, c->rs_l, (size_t)ws * hs))) return rc;
        const double fx = p->dense_scale, fy = p->dense_scale > 1.0 ? 1.0 : p->dense_scale;
        if ((rc = resize_inputs_dev(c, d_right, w, h, pitch, (uint8_t*)c->rs_r.p, ws, hs, fx, fy, c->stream)) ||
            (rc = resize_inputs_dev(c, d_left, w, h, pitch, (uint8_t*)c->rs_l.p, ws, hs, fx, fy, c->stream)))
            return rc;
        d_right = (const uint8_t*)c->rs_r.p; d_left = (const uint8_t*)c->rs_l.p;
        w = ws; h = hs; pitch = (size_t)ws;
    }
    SgmDims d;
    rc = make_dims(c, w, h, p, d);
    if (rc) return rc;

    const size_t npad = (size_t)d.Wp * d.h;
    const size_t vol = d.cells() * sizeof(uint16_t);
    if ((rc = ensure(c, c->bt1, npad * 8)) || (rc = ensure(c, c->bt2, (size_t)d.h * 6 * bt2_pitch(d.Wp) * 2 + 8192)) ||
        (rc = ensure(c, c->hsum, vol)) || (rc = ensure(c, c->C, vol)) ||
        (rc = ensure(c, c->S, vol)) ||
        (rc = ensure(c, c->sel_d16, (size_t)d.width1 * d.h * 2)) ||
        (rc = ensure(c, c->sel_key, (size_t)d.width1 * d.h * 4)) || (rc = ensure(c, c->raw, npad * 2)))
        return rc;

    hipStream_t s = c->stream;
    c->timings_valid = false;
    c->kev_n = 0;
    const int set = (int)(c->nsgm % wass_ctx::NSGM_SETS);
    c->ev = c->evs[set];
    WASS_HIP(c, hipEventRecord(c->ev[0], s));
    // wass_stereo.cpp:820-831: zero images, left at column D+off-comp, right at column D -- the padded pictures are not
    // materialised: the pre-filter reads the crops through the padding rule (and clears the frame's status words)
    if ((rc = launch_prefilter(c, d, d_right, d_left, pitch))) return rc;
    WASS_HIP(c, hipEventRecord(c->ev[1], s));
    if ((rc = launch_cost_volume(c, d))) return rc;
    WASS_HIP(c, hipEventRecord(c->ev[2], s));
    int nl = 0;
    if ((rc = launch_aggregate(c, d, &nl))) return rc;
    WASS_HIP(c, hipEventRecord(c->ev[3], s));
    if ((rc = launch_select(c, d))) return rc;
    WASS_HIP(c, hipEventRecord(c->ev[4], s));
    // with tail overlap the previous frame's clean-up may still be reading the caller's disparity buffer if the
    // caller does not alternate two of them; its first kernel is the only reader
    if (c->tail_overlap) WASS_HIP(c, hipStreamWaitEvent(s, c->ev_post, 0));
    if (d.speckle_win > 0) {
        // StereoSGBMImpl::compute: medianBlur, then filterSpeckles(disp, (minD - 1) * 16, window, 16 * range) on the whole padded
        // map (regions may reach into the columns that wass_stereo crops away), then the crop of :839
        if ((rc = ensure(c, c->raw2, npad * 2))) return rc;
        if ((rc = launch_median_full(c, d, (int16_t*)c->raw2.p))) return rc;
        if ((rc = speckle_filter_dev(c, (int16_t*)c->raw2.p, d.Wp, d.h, (d.minD - 1) * 16, d.speckle_win, 16 * d.speckle_range, s))) return rc;
        WASS_HIP(c, hipMemcpy2DAsync(d_out, (size_t)d.w * 2, (const int16_t*)c->raw2.p + d.D, (size_t)d.Wp * 2, (size_t)d.w * 2, d.h,
                                     hipMemcpyDeviceToDevice, s));
    } else if ((rc = launch_median_crop(c, d, d_out))) return rc;
    WASS_HIP(c, hipEventRecord(c->ev[5], s));
    // status word for wass_sgm_last_timings / the host entry point, in stream order (a blocking hipMemcpy on the
    // null stream would queue behind whatever else the process has in flight)
    WASS_HIP(c, hipMemcpyAsync(c->h_flags + 4 * set, c->flags.p, 4, hipMemcpyDeviceToHost, s));
    WASS_HIP(c, hipEventRecord(c->ev[6], s));
    c->last = d; c->have_last = true;
    c->launches[set] = nl;
    c->nsgm++;
    c->timings_valid = true;
    return WASS_OK;
}

// timings of SGM call number `call` (0-based), which must be one of the last two
static int read_timings(wass_ctx* c, unsigned long long call, wass_sgm_timings* out)
{
    const int set = (int)(call % wass_ctx::NSGM_SETS);
    hipEvent_t* ev = c->evs[set];
    WASS_HIP(c, hipEventSynchronize(ev[6]));
    wass_sgm_timings t = {};
    WASS_HIP(c, hipEventElapsedTime(&t.prefilter_ms, ev[0], ev[1]));
    WASS_HIP(c, hipEventElapsedTime(&t.cost_ms, ev[1], ev[2]));
    WASS_HIP(c, hipEventElapsedTime(&t.aggregate_ms, ev[2], ev[3]));
    WASS_HIP(c, hipEventElapsedTime(&t.select_ms, ev[3], ev[4]));
    WASS_HIP(c, hipEventElapsedTime(&t.median_ms, ev[4], ev[5]));
    WASS_HIP(c, hipEventElapsedTime(&t.total_ms, ev[0], ev[5]));
    WASS_HIP(c, hipEventElapsedTime(&t.vsum_ms, ev[7], ev[2]));
    const uint32_t fl = c->h_flags[4 * set];
    t.cost_overflow = (int)(fl & 1);
    t.aggregate_launches = c->launches[set];
    c->timings = t;
    *out = t;
    return WASS_OK;
}

int wass_sgm_last_timings(wass_ctx* c, wass_sgm_timings* out)
{
    if (!c || !out) return set_err(c, WASS_ERR_INVALID_ARG, "null argument");
    if (!c->timings_valid || c->nsgm == 0) return set_err(c, WASS_ERR_INVALID_ARG, "no completed wass_sgm_disparity call");
    return read_timings(c, c->nsgm - 1, out);
}

int wass_sgm_call_timings(wass_ctx* c, uint64_t call, wass_sgm_timings* out)
{
    if (!c || !out) return set_err(c, WASS_ERR_INVALID_ARG, "null argument");
    if (call == 0 || call > c->nsgm || c->nsgm - call >= (uint64_t)wass_ctx::NSGM_SETS)
        return set_err(c, WASS_ERR_INVALID_ARG, "the timings of call %llu are gone (calls so far: %llu, kept: the last %d)", (unsigned long long)call,
                       (unsigned long long)c->nsgm, wass_ctx::NSGM_SETS);
    return read_timings(c, call - 1, out);
}

int wass_sgm_call_count(wass_ctx* c, uint64_t* n_calls)
{
    if (!c || !n_calls) return set_err(c, WASS_ERR_INVALID_ARG, "null argument");
    *n_calls = c->nsgm;
    return WASS_OK;
}

int wass_sgm_prev_timings(wass_ctx* c, wass_sgm_timings* out)
{
    if (!c || !out) return set_err(c, WASS_ERR_INVALID_ARG, "null argument");
    if (c->nsgm < 2) return set_err(c, WASS_ERR_INVALID_ARG, "fewer than two wass_sgm_disparity calls so far");
    return read_timings(c, c->nsgm - 2, out);
}

int wass_ctx_set_kernel_events(wass_ctx* c, int on)
{
    if (!c) return WASS_ERR_INVALID_ARG;
    c->kernel_events = on != 0;
    return WASS_OK;
}

int wass_sgm_kernel_times(wass_ctx* c, char* names, size_t names_cap, float* ms, int max_kernels, int* n_kernels)
{
    if (!c || !names || !ms || !n_kernels || names_cap == 0) return set_err(c, WASS_ERR_INVALID_ARG, "null argument");
    if (!c->kernel_events || c->nsgm == 0) return set_err(c, WASS_ERR_INVALID_ARG, "no SGM call with wass_ctx_set_kernel_events(ctx, 1) so far");
    WASS_HIP(c, hipSetDevice(c->device));
    WASS_HIP(c, hipStreamSynchronize(c->stream));
    WASS_HIP(c, hipStreamSynchronize(c->side));
    std::string all;
    int n = 0;
    for (int i = 0; i < c->kev_n && n < max_kernels; ++i) {
        float t = 0;
        WASS_HIP(c, hipEventElapsedTime(&t, c->kev[i].a, c->kev[i].b));
        ms[n++] = t;
        all += c->kev[i].name;
        all += '\n';
    }
    if (all.size() + 1 > names_cap) return set_err(c, WASS_ERR_INVALID_ARG, "names buffer too small (%zu bytes needed)", all.size() + 1);
    memcpy(names, all.c_str(), all.size() + 1);
    *n_kernels = n;
    return WASS_OK;
}

int wass_sgm_selftest(wass_ctx* c, int w, int h, int num_disp, int ndirs, uint64_t* mismatches)
{
    if (!c || !mismatches) return set_err(c, WASS_ERR_INVALID_ARG, "null argument");
    if (w < 16 || h < 1 || w > 8192 || h > 8192) return set_err(c, WASS_ERR_INVALID_ARG, "bad self-test size");
    *mismatches = ~0ull;
    WASS_HIP(c, hipSetDevice(c->device));
    // a textured pair with a disparity ramp, from a fixed integer hash: bit-identical wherever this runs
    std::vector<uint8_t> R((size_t)w * h), L((size_t)w * h);
    auto tex = [](int x, int y) {
        uint32_t v = (uint32_t)(x >> 2) * 73856093u ^ (uint32_t)(y >> 1) * 19349663u;
        v ^= v >> 13; v *= 0x5bd1e995u; v ^= v >> 15;
        uint32_t n = (uint32_t)x * 2654435761u + (uint32_t)y * 40503u;
        n ^= n >> 16; n *= 0x85ebca6bu; n ^= n >> 13;
        return (int)(40 + (v % 150u) + (n % 24u));
    };
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            const int dsp = 2 + (num_disp * (y + 3 * x % 7)) / (3 * h + 9);      // inside (minD, D)
            L[(size_t)y * w + x] = (uint8_t)tex(x, y);
            R[(size_t)y * w + x] = (uint8_t)tex(x - dsp, y);
        }
    wass_sgm_params p;
    p.min_disp = 1; p.num_disp = num_disp; p.win = 5; p.P1 = 2 * 25; p.P2 = 64 * 25; p.uniq_ratio = 1; p.disp12_max_diff = -1;
    p.prefilter_cap = 60; p.speckle_win = -70; p.speckle_range = 16; p.ndirs = ndirs; p.disp_offset = 0; p.dense_scale = 1.0;
    std::vector<int16_t> disp((size_t)w * h);
    const bool was_debug = c->debug;
    c->debug = true;                                          // keep the finished S volume
    int rc = wass_sgm_disparity(c, R.data(), L.data(), w, h, (size_t)w, &p, disp.data());
    c->debug = was_debug;
    if (rc != WASS_OK && rc != WASS_ERR_COST_OVERFLOW) return rc;
    const SgmDims d = c->last;
    wass::Buf s2;
    unsigned long long* d_count = nullptr;
    if ((rc = ensure(c, s2, d.cells() * sizeof(uint16_t) + 256))) return rc;
    d_count = (unsigned long long*)((char*)s2.p + d.cells() * sizeof(uint16_t));
    d_count = (unsigned long long*)(((uintptr_t)d_count + 7) & ~(uintptr_t)7);
    hipError_t e = hipMemsetAsync(d_count, 0, 8, c->stream);
    if (e == hipSuccess) rc = selftest_reference(c, d, (uint32_t*)s2.p, d_count);
    unsigned long long hc = ~0ull;
    if (e == hipSuccess && rc == WASS_OK) e = hipMemcpyAsync(&hc, d_count, 8, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    (void)hipFree(s2.p);
    if (rc != WASS_OK) return rc;
    if (e != hipSuccess) return set_err(c, WASS_ERR_DEVICE, "self-test: %s", hipGetErrorString(e));
    *mismatches = hc;
    if (hc) return set_err(c, WASS_ERR_DEVICE, "self-test: the production schedule and the plain per-path sweeps disagree in %llu cells of S "
                                              "(%dx%d, D=%d, %d paths)", hc, w, h, num_disp, ndirs);
    return WASS_OK;
}

int wass_sgm_probe_vsum(wass_ctx* c, float* plain_ms, float* production_ms)
{
    if (!c || !plain_ms || !production_ms) return set_err(c, WASS_ERR_INVALID_ARG, "null argument");
    if (!c->have_last) return set_err(c, WASS_ERR_INVALID_ARG, "no completed wass_sgm_disparity call");
    WASS_HIP(c, hipSetDevice(c->device));
    WASS_HIP(c, hipStreamSynchronize(c->stream));
    WASS_HIP(c, hipStreamSynchronize(c->side));
    WASS_HIP(c, hipStreamSynchronize(c->tail));
    hipEvent_t e[3] = { nullptr, nullptr, nullptr };
    for (auto& x : e) if (hipEventCreate(&x) != hipSuccess) { for (auto& y : e) if (y) (void)hipEventDestroy(y); return set_err(c, WASS_ERR_DEVICE, "hipEventCreate failed"); }
    int rc = WASS_OK;
    float best[2] = { 1e30f, 1e30f };
    for (int rep = 0; rep < 3 && rc == WASS_OK; ++rep) {        // the hsum volume of the last frame is still there; both forms write the same C
        (void)hipEventRecord(e[0], c->stream);
        rc = launch_vsum_only(c, c->last, true);
        (void)hipEventRecord(e[1], c->stream);
        if (rc == WASS_OK) rc = launch_vsum_only(c, c->last, false);
        (void)hipEventRecord(e[2], c->stream);
        if (hipStreamSynchronize(c->stream) != hipSuccess) rc = set_err(c, WASS_ERR_DEVICE, "probe failed");
        float a = 0, b = 0;
        if (rc == WASS_OK && hipEventElapsedTime(&a, e[0], e[1]) == hipSuccess && hipEventElapsedTime(&b, e[1], e[2]) == hipSuccess) {
            best[0] = a < best[0] ? a : best[0];
            best[1] = b < best[1] ? b : best[1];
        }
    }
    for (auto& x : e) (void)hipEventDestroy(x);
    if (rc != WASS_OK) return rc;
    *plain_ms = best[0]; *production_ms = best[1];
    return WASS_OK;
}

int wass_sgm_disparity(wass_ctx* c, const uint8_t* right, const uint8_t* left, int w, int h, size_t pitch,
                       const wass_sgm_params* p, int16_t* disp16_out)
{
    if (!c || !right || !left || !disp16_out) return set_err(c, WASS_ERR_INVALID_ARG, "null argument");
    if (w <= 0 || h <= 0 || pitch < (size_t)w) return set_err(c, WASS_ERR_INVALID_ARG, "bad image geometry");
    WASS_HIP(c, hipSetDevice(c->device));
    const size_t n = (size_t)w * h;
    int rc, ws = w, hs = h;
    if (p && p->dense_scale != 1.0 && wass_dense_input_size(w, h, p->dense_scale, &ws, &hs) != WASS_OK)
        return set_err(c, WASS_ERR_INVALID_ARG, "DENSE_SCALE %g leaves no image", p->dense_scale);
    const size_t no = (size_t)ws * hs;                                   // the map has the size of the RESIZED inputs
    if ((rc = ensure(c, c->tmp_in0, n)) || (rc = ensure(c, c->tmp_in1, n)) || (rc = ensure(c, c->tmp_out, no * 2)))
        return rc;
    WASS_HIP(c, hipMemcpy2DAsync(c->tmp_in0.p, w, right, pitch, w, h, hipMemcpyHostToDevice, c->stream));
    WASS_HIP(c, hipMemcpy2DAsync(c->tmp_in1.p, w, left, pitch, w, h, hipMemcpyHostToDevice, c->stream));
    rc = wass_sgm_disparity_dev(c, (const uint8_t*)c->tmp_in0.p, (const uint8_t*)c->tmp_in1.p, w, h, w, p,
                                (int16_t*)c->tmp_out.p);
    if (rc) return rc;
    WASS_HIP(c, hipMemcpyAsync(disp16_out, c->tmp_out.p, no * 2, hipMemcpyDeviceToHost, c->stream));
    WASS_HIP(c, hipStreamSynchronize(c->stream));
    const uint32_t fl = c->h_flags[4 * (int)((c->nsgm - 1) % wass_ctx::NSGM_SETS)];
    if (fl & 1)
        return set_err(c, WASS_ERR_COST_OVERFLOW,
                       "block cost + P2 exceeded 32767: outside the range where the reference is well defined");
    return WASS_OK;
}

int wass_sgm_debug_fetch(wass_ctx* c, int16_t* C_out, int16_t* S_out, int16_t* raw_out)
{
    if (!c) return WASS_ERR_INVALID_ARG;
    if (!c->have_last) return set_err(c, WASS_ERR_INVALID_ARG, "no completed wass_sgm_disparity call");
    if (S_out && !c->debug)
        return set_err(c, WASS_ERR_INVALID_ARG, "S is only kept when wass_ctx_set_debug(ctx, 1) was set before the call");
    const SgmDims& d = c->last;
    WASS_HIP(c, hipStreamSynchronize(c->stream));
    // volumes are [h][width1][Dp] on the device; the caller gets [h][width1][D]
    const size_t npx = (size_t)d.h * d.width1;
    if (C_out)
        WASS_HIP(c, hipMemcpy2D(C_out, (size_t)d.D * 2, c->C.p, (size_t)d.Dp * 2, (size_t)d.D * 2, npx, hipMemcpyDeviceToHost));
    if (S_out)
        WASS_HIP(c, hipMemcpy2D(S_out, (size_t)d.D * 2, c->S.p, (size_t)d.Dp * 2, (size_t)d.D * 2, npx, hipMemcpyDeviceToHost));
    if (raw_out)
        WASS_HIP(c, hipMemcpy(raw_out, c->raw.p, (size_t)d.Wp * d.h * 2, hipMemcpyDeviceToHost));
    return WASS_OK;
}

}  // extern "C"
