// abi_guard.hpp -- exception barrier of the C ABI.  include/hipvoc.h promises that no function aborts or throws across
// the boundary (the Rust host would see undefined behaviour; the worker must answer an error response and keep serving,
// src/worker/mod.rs:246-258).  Every extern "C" body that can allocate runs inside abi_guard(): std::bad_alloc becomes
// HIPVOC_E_NOMEM, any other exception HIPVOC_E_INTERNAL, with the text handed to `on_error` (which must not throw).
// Pure C++ (no HIP): tests/native/test_abi_guard.cpp exercises it.
#pragma once
#include <exception>
#include <new>

#ifndef HIPVOC_E_NOMEM
#define HIPVOC_E_NOMEM (-8)
#define HIPVOC_E_INTERNAL (-9)
#endif

namespace hipvoc {

template <typename Body, typename OnError>
static inline int abi_guard(Body &&body, OnError &&on_error) noexcept {
    try {
        return body();
    } catch (const std::bad_alloc &) {
        on_error("out of host memory");
        return HIPVOC_E_NOMEM;
    } catch (const std::exception &e) {
        on_error(e.what());
        return HIPVOC_E_INTERNAL;
    } catch (...) {
        on_error("unknown C++ exception");
        return HIPVOC_E_INTERNAL;
    }
}

}  // namespace hipvoc
