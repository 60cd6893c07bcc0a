// device code + launch stub of ONE XCD-fused kernel instance (dispatch.hpp launch_xcd_sel); built once per instance id
// (Makefile: -DMI355_XCD_ID=k), see the note in dispatch.hpp on why every instance has its own translation unit
#define MI355_XCD_DEFINE_INSTANCES
#include "hip_launcher.hpp"
#ifndef MI355_XCD_ID
#error "compile with -DMI355_XCD_ID=<instance id>"
#endif
namespace mi355 {
static_assert(MI355_XCD_ID >= 0 && MI355_XCD_ID < XCD_INSTANCE_COUNT, "no such XCD kernel instance: update XCD_IDS in the Makefile");
template bool launch_xcd_sel<MI355_XCD_ID, HipLauncher>(int, const XcdFusedArgs&, unsigned, HipLauncher&);
}
