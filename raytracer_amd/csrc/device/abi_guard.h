// No exception crosses the C boundary of librt_mi355.so (include/rt_mi355.h: "nothing throws
// across this boundary"): every extern "C" function of the library is a function-try-block
// closed by one of the macros below, so that a std::bad_alloc from a container, a
// std::system_error from a thread that could not be started or anything else becomes a
// status code.  tests/test_abi.py checks that no exported definition is left without one.
#pragma once
#include <new>

#include "rt_mi355.h"

// Record the failure on the handle (best effort) and return its status; defined next to the
// struct each one writes to.  They never throw.
int rt_abi_caught(RtContext* ctx, bool out_of_memory) noexcept;
int rt_abi_caught(RtGroup* group, bool out_of_memory) noexcept;

#define RT_ABI_END_CTX(handle)                                            \
    catch (const std::bad_alloc&) { return rt_abi_caught(handle, true); } \
    catch (...) { return rt_abi_caught(handle, false); }
#define RT_ABI_END_INT                                  \
    catch (const std::bad_alloc&) { return RT_ERR_OOM; } \
    catch (...) { return RT_ERR_HIP; }
#define RT_ABI_END_VOID catch (...) {}
#define RT_ABI_END_VALUE(value) catch (...) { return value; }
