// Shared host-side helpers of libasep_hip.so (error reporting, weight-blob parsing, device buffers).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <new>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/asep_hip.h"

namespace asep {

void set_error(const char* fmt, ...);
const char* get_error();
void warn_ignored_switches();   // stderr, once per process: ASEP_* switches of earlier rounds that this build no longer reads

#define ASEP_HIP_CHECK(expr)                                                                    \
    do {                                                                                        \
        hipError_t _e = (expr);                                                                 \
        if (_e != hipSuccess) {                                                                 \
            asep::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__,    \
                            __LINE__);                                                          \
            return ASEP_ERR_HIP;                                                                \
        }                                                                                       \
    } while (0)

#define ASEP_HIP_CHECK_THROW(expr)                                                              \
    do {                                                                                        \
        hipError_t _e = (expr);                                                                 \
        if (_e != hipSuccess) {                                                                 \
            asep::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__,    \
                            __LINE__);                                                          \
            throw asep::HipError();                                                             \
        }                                                                                       \
    } while (0)

struct HipError {};
struct ArgError {};

// Every extern "C" entry runs its body between these two: no C++ exception (HipError / ArgError of the helpers,
// std::bad_alloc or std::length_error of the std::vector / std::map / std::string code) crosses the C ABI.
#define ASEP_GUARD_BEGIN try {
#define ASEP_GUARD_END_WITH(HIP_RC, ARG_RC)                                                     \
    }                                                                                           \
    catch (const asep::HipError&) { return HIP_RC; }                                            \
    catch (const asep::ArgError&) { return ARG_RC; }                                            \
    catch (const std::bad_alloc&) {                                                             \
        asep::set_error("%s: out of host memory", __func__);                                    \
        return HIP_RC;                                                                          \
    }                                                                                           \
    catch (const std::exception& e) {                                                           \
        asep::set_error("%s: exception: %s", __func__, e.what());                               \
        return HIP_RC;                                                                          \
    }                                                                                           \
    catch (...) {                                                                               \
        asep::set_error("%s: unknown exception", __func__);                                     \
        return HIP_RC;                                                                          \
    }
#define ASEP_GUARD_END ASEP_GUARD_END_WITH(ASEP_ERR_HIP, ASEP_ERR_ARG)
#define ASEP_GUARD_END_PTR ASEP_GUARD_END_WITH(nullptr, nullptr)

struct HostTensor {
    std::vector<int> dims;
    std::vector<float> data;       // copied out of the caller's blob (payloads may be unaligned)
    size_t count() const {
        size_t n = 1;
        for (int d : dims) n *= (size_t)d;
        return n;
    }
};

// Parses the "ASEPW001" container (weights.py).  Returns false (and sets the error) on malformed input.
bool parse_blob(const void* blob, size_t nbytes, std::map<std::string, HostTensor>& out);

// Buffers requested in a deterministic order by a forward pass; re-used across calls.
class BufferPool {
public:
    ~BufferPool() { release(); }
    void begin() { next_ = 0; }
    void* get(size_t bytes);   // throws HipError
    void release();
    size_t total_bytes() const;
private:
    struct Buf { void* p; size_t n; };
    std::vector<Buf> bufs_;
    size_t next_ = 0;
};

inline int cdiv(int a, int b) { return (a + b - 1) / b; }

// Device pointer + {H, W, C} of a named end point of the last ARU forward (aru_engine.hip); library-internal.
int aru_endpoint_dev(asep_aru* m, const char* name, const float** d_ptr, int dims[3], int* is_bf16 = nullptr);
// Number of channels the end point `name` will have for this model's configuration, or -1 if unknown.
int aru_endpoint_channels(const asep_aru* m, const char* name);
// Number of output classes (channels of the logits / probability map) of the model.
int aru_num_classes(const asep_aru* m);

}  // namespace asep
