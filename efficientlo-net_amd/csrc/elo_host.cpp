// elo_host.cpp -- host-side plumbing of the C ABI (errors, version).
#include "elo_common.h"

namespace elo {

char *err_buf()
{
    static thread_local char buf[512] = "";
    return buf;
}

int fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(err_buf(), 512, fmt, ap);
    va_end(ap);
    return code;
}

int check_launch(const char *what)
{
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(ELO_ERR_LAUNCH, "%s: %s", what, hipGetErrorString(e));
    return ELO_OK;
}

static const elo_tuning kDefaults = {/*chain_forms*/ 1, /*narrow_mfma*/ 1, /*range_check*/ 0, /*select_dense_waves*/ 0, /*random_dense_rows*/ 0,
                                     /*setconv_chain_rows*/ -1, /*mlp_chain_rows*/ -1, /*small_tile_units*/ 64,
                                     /*pool_wave*/ 1};
elo_tuning &tuning()            // what the launchers read: elo_set_tuning's value with the elo_debug_* overrides on top
{
    static elo_tuning t = kDefaults;
    return t;
}
elo_tuning &tuning_base()       // what elo_set_tuning installed: an elo_debug_*(-1) call puts its field back to this
{
    static elo_tuning t = kDefaults;
    return t;
}

}  // namespace elo

extern "C" int elo_get_tuning(elo_tuning *out)
{
    if (!out) return elo::fail(ELO_ERR_ARG, "elo_get_tuning: null pointer");
    *out = elo::tuning();
    return ELO_OK;
}

extern "C" int elo_get_tuning_base(elo_tuning *out)
{
    if (!out) return elo::fail(ELO_ERR_ARG, "elo_get_tuning_base: null pointer");
    *out = elo::tuning_base();
    return ELO_OK;
}

extern "C" int elo_set_tuning(const elo_tuning *in)
{
    const char *who = "elo_set_tuning";
    if (!in) return elo::fail(ELO_ERR_ARG, "%s: null pointer", who);
    if ((in->chain_forms | 1) != 1 || (in->narrow_mfma | 1) != 1 || (in->range_check | 1) != 1 || (in->pool_wave | 1) != 1)
        return elo::fail(ELO_ERR_ARG, "%s: chain_forms / narrow_mfma / range_check / pool_wave are 0 or 1", who);
    if (in->select_dense_waves != 0 && in->select_dense_waves != 4 && in->select_dense_waves != 8 && in->select_dense_waves != 16)
        return elo::fail(ELO_ERR_ARG, "%s: select_dense_waves is 0, 4, 8 or 16", who);
    if (in->random_dense_rows != 0 && in->random_dense_rows != 2 && in->random_dense_rows != 4)
        return elo::fail(ELO_ERR_ARG, "%s: random_dense_rows is 0, 2 or 4", who);
    if (in->setconv_chain_rows < -1 || in->mlp_chain_rows < -1 || in->small_tile_units < 0)
        return elo::fail(ELO_ERR_ARG, "%s: row thresholds are -1 (the regime's default) or >= 0", who);
    elo::tuning() = *in;
    elo::tuning_base() = *in;
    return ELO_OK;
}

// One lane submit of the host runtime as ONE native call: (ordering against the producer of the input,) the optional device-to-device
// copy of the lane's input, then the lane's instantiated graph, all on the lane's stream (include/elo.h).  Through torch the same
// sequence -- Event.record + Stream.wait_event + Tensor.copy_ + CUDAGraph.replay() -- costs the submitting thread ~35 us a step, this
// 16-18 (tools/submit_native_probe.py): with 20 steps between two synchronisations (the driver's protocol) the four queues' first
// forwards start that much closer together.
extern "C" int elo_graph_submit(void *graph_exec, elo_stream_t stream, void *dst, const void *src, unsigned long bytes,
                                elo_stream_t producer, void *order_event, int device)
{
    const char *who = "elo_graph_submit";
    if (!graph_exec) return elo::fail(ELO_ERR_ARG, "%s: null graph", who);
    if (bytes && (!dst || !src)) return elo::fail(ELO_ERR_ARG, "%s: a copy needs both pointers", who);
    hipStream_t s = (hipStream_t)stream;
    int restore = -1;
    if (device >= 0) {                      // the lane's device becomes the calling thread's for the duration of the call
        int cur = -1;
        if (hipGetDevice(&cur) != hipSuccess) return elo::fail(ELO_ERR_LAUNCH, "%s: hipGetDevice failed", who);
        if (cur != device) {
            if (hipSetDevice(device) != hipSuccess) return elo::fail(ELO_ERR_ARG, "%s: hipSetDevice(%d) failed", who, device);
            restore = cur;
        }
    }
    hipError_t e = hipSuccess;
    const char *what = "";
    // the lane's stream runs nothing of this step before the producer's work so far -- an IDLE producer has none (hipStreamQuery:
    // one host call, no packet): the event pair would put a cross-queue barrier into the lane's chain for nothing (measured: 10 % of
    // the saturated batch-1 rate when every step carries one)
    if (order_event && (hipStream_t)producer != s && hipStreamQuery((hipStream_t)producer) != hipSuccess) {
        (void)hipGetLastError();                          // (hipErrorNotReady is the answer, not an error)
        what = "hipEventRecord";
        e = hipEventRecord((hipEvent_t)order_event, (hipStream_t)producer);
        if (e == hipSuccess) { what = "hipStreamWaitEvent"; e = hipStreamWaitEvent(s, (hipEvent_t)order_event, 0); }
    }
    if (e == hipSuccess && bytes) { what = "hipMemcpyAsync"; e = hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, s); }
    if (e == hipSuccess) { what = "hipGraphLaunch"; e = hipGraphLaunch((hipGraphExec_t)graph_exec, s); }
    if (restore >= 0) (void)hipSetDevice(restore);
    if (e != hipSuccess) return elo::fail(ELO_ERR_LAUNCH, "%s: %s: %s", who, what, hipGetErrorString(e));
    return ELO_OK;
}

extern "C" int elo_abi_version(void) { return 26; }
extern "C" const char *elo_last_error(void) { return elo::err_buf(); }
