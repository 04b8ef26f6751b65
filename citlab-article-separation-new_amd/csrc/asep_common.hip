// Runtime entry points and shared helpers of libasep_hip.so.
#include "asep_common.h"
#include <cstdlib>

namespace asep {

static thread_local char g_err[1024] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
const char* get_error() { return g_err; }

// Engine switches of earlier rounds whose experiments left the tree, and the two that only an ABLATION build reads:
// a script or a log that still sets one of them measures the DEFAULT path.  Said once per process on stderr when a model is loaded.
void warn_ignored_switches() {
    static bool done = false;
    if (done) return;
    done = true;
    static const char* const gone[] = {
        "ASEP_F32_SPLIT", "ASEP_SPLIT_L0", "ASEP_SPLIT_ALDS", "ASEP_SPLIT_TH16", "ASEP_WINOGRAD", "ASEP_WINO_REG", "ASEP_WINO16", "ASEP_BIGTILE",
        "ASEP_BIGTILE2", "ASEP_XCD_ONESHOT", "ASEP_SIDE_STREAM", "ASEP_FUSED8_VAR", "ASEP_BF_TH8", "ASEP_BF_MTB", "ASEP_BF_W8", "ASEP_BF_R8B",
        "ASEP_BF_R8F", "ASEP_CONVS_DBG", "ASEP_R8S_DBG", "ASEP_MAXP24",
#ifndef ASEP_ABLATION
        "ASEP_GNN_BATCH", "ASEP_GNN_LANES",
#endif
    };
    for (const char* name : gone)
        if (getenv(name))
            fprintf(stderr, "libasep_hip: %s is set but is not a switch of this build (ignored; asep_engine_switches() lists what is read)\n", name);
}

bool parse_blob(const void* blob, size_t nbytes, std::map<std::string, HostTensor>& out) {
    const uint8_t* p = (const uint8_t*)blob;
    if (!p || nbytes < 12 || memcmp(p, "ASEPW001", 8) != 0) {
        set_error("weight blob: bad magic (expected ASEPW001)");
        return false;
    }
    uint32_t n;
    memcpy(&n, p + 8, 4);
    size_t off = 12;
    for (uint32_t i = 0; i < n; ++i) {
        if (off + 2 > nbytes) { set_error("weight blob: truncated"); return false; }
        uint16_t nl;
        memcpy(&nl, p + off, 2);
        off += 2;
        if (off + nl + 1 > nbytes) { set_error("weight blob: truncated"); return false; }
        std::string name((const char*)p + off, nl);
        off += nl;
        uint8_t nd = p[off];
        off += 1;
        HostTensor t;
        if (off + 4u * nd > nbytes) { set_error("weight blob: truncated"); return false; }
        for (int d = 0; d < nd; ++d) {
            uint32_t v;
            memcpy(&v, p + off, 4);
            off += 4;
            t.dims.push_back((int)v);
        }
        size_t cnt = t.count();
        if (off + 4 * cnt > nbytes) { set_error("weight blob: truncated in %s", name.c_str()); return false; }
        t.data.resize(cnt);
        memcpy(t.data.data(), p + off, 4 * cnt);
        off += 4 * cnt;
        out[name] = std::move(t);
    }
    if (off != nbytes) { set_error("weight blob: %zu trailing bytes", nbytes - off); return false; }
    return true;
}

void* BufferPool::get(size_t bytes) {
    if (bytes == 0) bytes = 16;
    if (next_ < bufs_.size()) {
        Buf& b = bufs_[next_];
        if (b.n < bytes) {
            if (b.p) (void)hipFree(b.p);
            b.p = nullptr;
            b.n = 0;
            ASEP_HIP_CHECK_THROW(hipMalloc(&b.p, bytes));
            b.n = bytes;
        }
        ++next_;
        return b.p;
    }
    Buf b{nullptr, 0};
    ASEP_HIP_CHECK_THROW(hipMalloc(&b.p, bytes));
    b.n = bytes;
    bufs_.push_back(b);
    ++next_;
    return b.p;
}

void BufferPool::release() {
    for (auto& b : bufs_)
        if (b.p) (void)hipFree(b.p);
    bufs_.clear();
    next_ = 0;
}

size_t BufferPool::total_bytes() const {
    size_t t = 0;
    for (auto& b : bufs_) t += b.n;
    return t;
}

}  // namespace asep

extern "C" {

const char* asep_last_error(void) { return asep::get_error(); }
const char* asep_version(void) { return "asep_hip 0.5 (gfx950)"; }

// The environment switches this BUILD reads, one name per line.
const char* asep_engine_switches(void) {
    return "ASEP_FUSE_POOL\nASEP_FUSE_ACT\nASEP_C12\nASEP_FUSED8\nASEP_R8_VALU\nASEP_XCD_SCHED\nASEP_BF_RES32\nASEP_BF_WALK\nASEP_BF_CONVR\nASEP_SPLIT_DECONV\nASEP_LANES\nASEP_GNN_STEP\nASEP_GNN_FACTOR"
#ifdef ASEP_ABLATION
           "\nASEP_GNN_BATCH\nASEP_GNN_LANES"
#endif
        ;
}
int asep_abi_version(void) { return ASEP_ABI_VERSION; }

int asep_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int asep_init(int device_id) {
    int n = 0;
    ASEP_HIP_CHECK(hipGetDeviceCount(&n));
    if (device_id < 0 || device_id >= n) {
        asep::set_error("asep_init: device %d out of range (%d devices)", device_id, n);
        return ASEP_ERR_ARG;
    }
    ASEP_HIP_CHECK(hipSetDevice(device_id));
    hipDeviceProp_t prop;
    ASEP_HIP_CHECK(hipGetDeviceProperties(&prop, device_id));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        asep::set_error("asep_init: device %d is %s; this library is built for gfx950 only", device_id,
                        prop.gcnArchName);
        return ASEP_ERR_UNSUPPORTED;
    }
    return ASEP_OK;
}

// ---- page-locked host memory: what makes the host-pointer entry points and the decode slots DMA-able ------------
void* asep_host_alloc(size_t nbytes) {
    void* p = nullptr;
    if (nbytes == 0 || hipHostMalloc(&p, nbytes, hipHostMallocDefault) != hipSuccess) {
        asep::set_error("asep_host_alloc: cannot page-lock %zu bytes", nbytes);
        return nullptr;
    }
    return p;
}

void asep_host_free(void* p) {
    if (p) (void)hipHostFree(p);
}

int asep_host_register(void* p, size_t nbytes) {
    if (!p || nbytes == 0) { asep::set_error("asep_host_register: bad argument"); return ASEP_ERR_ARG; }
    ASEP_HIP_CHECK(hipHostRegister(p, nbytes, hipHostRegisterDefault));
    return ASEP_OK;
}

int asep_host_unregister(void* p) {
    if (!p) return ASEP_ERR_ARG;
    ASEP_HIP_CHECK(hipHostUnregister(p));
    return ASEP_OK;
}

}  // extern "C"
