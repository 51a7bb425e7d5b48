// abi_guard.hpp -- nothing is thrown across the C ABI (include/raytracer_amd.h, "Conventions").
//
// Every extern "C" entry point is a function-try-block whose handler calls abi_exception(): the exception in flight
// is classified (std::bad_alloc and thread-creation failures -> RT_E_NOMEM, anything else -> RT_E_INVALID) and its
// text is kept for rt_last_error.  The reference catches std::exception once in main (src/main.cpp:8-14); the C++
// wrapper raytracer_amd/host/raytracer.hpp turns the code back into std::runtime_error there.
#pragma once
#include <cstdio>
#include <exception>
#include <new>
#include <string>
#include <system_error>

#include "../../include/raytracer_amd.h"

namespace rt {

// Call from inside a catch (...) handler only.  `err` (may be null) receives the message; never throws.
inline int abi_exception(std::string* err) noexcept {
    int code = RT_E_INVALID;
    const char* what = "unknown C++ exception";
    char buf[256];
    try {
        throw;
    } catch (const std::bad_alloc&) {
        code = RT_E_NOMEM; what = "out of host memory";
    } catch (const std::system_error& e) {          // std::thread could not be started (EAGAIN) and the like
        code = RT_E_NOMEM;
        std::snprintf(buf, sizeof buf, "system resource error: %s", e.what());
        what = buf;
    } catch (const std::exception& e) {
        std::snprintf(buf, sizeof buf, "%s", e.what());
        what = buf;
    } catch (...) {
    }
    if (err) {
        try { err->assign(what); } catch (...) { err->clear(); }     // clear() does not allocate
    }
    return code;
}

}  // namespace rt
