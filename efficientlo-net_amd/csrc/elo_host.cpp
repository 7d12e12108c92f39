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

}  // namespace elo

extern "C" int elo_abi_version(void) { return 19; }
extern "C" const char *elo_last_error(void) { return elo::err_buf(); }
