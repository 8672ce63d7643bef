// rf_host.hpp -- host-side error reporting shared by the translation units of libradfoam_hip.so.
//
// The reference reports every failure by throwing std::runtime_error (pipeline_bindings.cpp:14-70,
// cuda_helpers.h:12-19).  The C-ABI returns a status and keeps the message in a thread-local
// buffer that rf_last_error() hands out.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdio>

#include "../../include/radfoam_hip.h"

namespace rf {

inline thread_local char g_err[512] = "";

inline int fail(int code, const char *fmt, const char *detail = "") {
    std::snprintf(g_err, sizeof(g_err), fmt, detail);
    return code;
}

inline int check_launch(const char *what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        std::snprintf(g_err, sizeof(g_err), "%s: %s", what, hipGetErrorString(e));
        return RF_ERR_LAUNCH;
    }
    return RF_OK;
}

}  // namespace rf
