// api.hip — extern "C" surface of libnbci.so (see include/nbci.h).
#include "nbci_common.h"
#include "../../include/nbci.h"

namespace nbci {
static thread_local std::string g_last_error;
void set_error(const std::string& msg) { g_last_error = msg; }
int fail(int code, const std::string& msg) { g_last_error = msg; return code; }
int gemm_launch(const nbci_gemm_desc& d, hipStream_t stream);
}  // namespace nbci

extern "C" {

int nbci_version(void) { return NBCI_VERSION; }
const char* nbci_last_error(void) { return nbci::g_last_error.c_str(); }

int nbci_gemm(const nbci_gemm_desc* d, nbci_stream_t stream) {
    if (!d) return nbci::fail(NBCI_EINVAL, "nbci_gemm: null desc");
    return nbci::gemm_launch(*d, (hipStream_t)stream);
}

}  // extern "C"
