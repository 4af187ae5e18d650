// Error plumbing and version of libtensoflow_hip.so (no exceptions cross the C ABI).
#include <stdarg.h>
#include <stdio.h>

#include "../../include/tensoflow_hip.h"
#include "tf_internal.h"

static thread_local char g_err[512] = "";

void tf_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* tf_last_error(void) { return g_err; }
extern "C" int tf_version(void) { return 100; /* 0.1.0 */ }

static thread_local TfLaunchBudget g_budget;
const TfLaunchBudget& tf_launch_budget() { return g_budget; }
extern "C" int tf_set_launch_budget(int32_t bvh_blocks_per_cu, int32_t flow_waves_per_block, int32_t inner_teams) {
  TF_REQUIRE(bvh_blocks_per_cu >= 0 && bvh_blocks_per_cu <= 8, TF_EINVAL, "tf_set_launch_budget: bvh_blocks_per_cu must be 0..8");
  TF_REQUIRE(flow_waves_per_block == 0 || flow_waves_per_block == 4 || flow_waves_per_block == 8 || flow_waves_per_block == 12, TF_EINVAL,
             "tf_set_launch_budget: flow_waves_per_block must be 0, 4, 8 or 12");
  TF_REQUIRE(inner_teams >= 0 && inner_teams <= 2, TF_EINVAL, "tf_set_launch_budget: inner_teams must be 0, 1 or 2");
  g_budget.bvh_blocks_per_cu = bvh_blocks_per_cu;
  g_budget.flow_waves_per_block = flow_waves_per_block;
  g_budget.inner_teams = inner_teams;
  return TF_OK;
}
