// anm_mpc_only.hip -- the MPC kernel + its part of the C ABI (include/anm_mi355x.h: anm_mpc_*) for ONE network topology,
// without the step kernels.  Networks above 12 buses step on the table-driven lane-group kernels of any built library
// (no per-topology compile); the MPC kernel is specialised on the topology's sizes, so those networks get this small
// library (libmpc_<topology>.so, gym_anm_amd/codegen.py: build_library(..., mpc_only=True)) when an MPC agent is made.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include ANM_TOPO_HEADER
#include "anm_pack.hpp"
#include "anm_mpc.hpp"

using namespace anm;

#include "anm_mpc_capi_types.inc"

namespace {

thread_local std::string g_err;

int fail(const char* what) {
  g_err = what;
  return -1;
}
int fail_hip(hipError_t e, const char* where) {
  g_err = std::string(where) + ": " + hipGetErrorString(e);
  return -2;
}

}  // namespace

extern "C" {

const char* anm_last_error(void) { return g_err.c_str(); }

const char* anm_topology_signature(void) { return Topo::SIGNATURE; }

#include "anm_mpc_capi.inc"

}  // extern "C"
