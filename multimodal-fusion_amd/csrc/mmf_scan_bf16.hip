// mmf_scan_bf16.hip — placeholder until the bf16 MFMA scan lands (this file is replaced next).
#include "mmf_host.h"
namespace mmf {
int scan_bf16_supported(int64_t, int, int) { return 0; }
}  // namespace mmf
