// Version and error strings of the C ABI (include/dvsof.h).
#include "common.h"

extern "C" {

int dvsof_version(void) { return DVSOF_VERSION; }

const char *dvsof_error_string(int code)
{
    switch (code) {
    case DVSOF_OK: return "ok";
    case DVSOF_EINVAL: return "invalid argument (shape, null pointer or unsupported size)";
    case DVSOF_ENOSPACE: return "workspace too small";
    case DVSOF_ECOMM: return "RCCL is missing or a communicator / collective call failed";
    default: break;
    }
    if (code > 0) return hipGetErrorString((hipError_t)code);
    return "unknown dvsof error";
}

}  // extern "C"
