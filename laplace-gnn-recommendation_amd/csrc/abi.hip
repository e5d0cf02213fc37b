// ABI bookkeeping for include/laplace_hip.h.
#include "common.hpp"

extern "C" {

int mi_abi_version(void) { return MI_ABI_VERSION; }

const char* mi_error_string(int code) {
    switch (code) {
        case 0: return "success";
        case MI_ERR_BAD_ARG: return "bad argument (null pointer, negative size or misaligned buffer)";
        case MI_ERR_TOO_LARGE: return "size does not fit 32-bit indexing";
        case MI_ERR_WORKSPACE: return "workspace too small";
        case MI_ERR_UNSUPPORTED: return "unsupported feature width or k";
        default: break;
    }
    if (code > 0) return hipGetErrorString((hipError_t)code);
    return "unknown error";
}

}  // extern "C"
