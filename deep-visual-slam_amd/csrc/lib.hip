// Library-level entry points of libdvslam_hip.so (include/dvslam.h: "library").
#include "common.h"

#include <cstring>

namespace dvs {

char* err_buf() {
    static thread_local char buf[512] = {0};
    return buf;
}

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(err_buf(), 512, fmt, ap);
    va_end(ap);
    return code;
}

}  // namespace dvs

extern "C" {

const char* dvs_last_error(void) { return dvs::err_buf(); }

int dvs_abi_version(void) { return 1; }

const char* dvs_arch(void) { return "gfx950"; }

}  // extern "C"
