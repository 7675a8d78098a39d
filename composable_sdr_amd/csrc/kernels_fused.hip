// Fused channelizer kernels (placeholder until the first fused kernel lands).
#include "fused.h"
namespace csdr {
struct FusedPlan {};
bool fused_supported(uint32_t, uint32_t) { return false; }
int fused_create(const FusedConfig &, FusedPlan **) { set_error("no fused kernel"); return -1; }
int fused_reset(FusedPlan *, hipStream_t) { return 0; }
int fused_process(FusedPlan *, const FusedCall &, hipStream_t) { set_error("no fused kernel"); return -1; }
const char *fused_name(const FusedPlan *) { return "none"; }
void fused_destroy(FusedPlan *) {}
}
