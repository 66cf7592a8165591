// Stream plumbing for the host side: ordering between two HIP streams without torch objects.
#include "common.h"
#include "../../include/speecht5_hip.h"

namespace {
constexpr int NEV = 64;
hipEvent_t g_ev[NEV];
bool g_ev_init = false;
int g_ev_next = 0;
}  // namespace

/* Everything enqueued on `to` after this call runs after everything enqueued on `from` before it (event record on `from`,
 * wait on `to`; events come from a small ring -- a wait captures the record that precedes it, so re-recording an event
 * that an earlier wait still refers to is legal). */
extern "C" int st5_stream_fork(void* from, void* to) {
  if (!g_ev_init) {
    for (int i = 0; i < NEV; ++i)
      if (hipEventCreateWithFlags(&g_ev[i], hipEventDisableTiming) != hipSuccess) return ST5_ERR_LAUNCH;
    g_ev_init = true;
  }
  hipEvent_t ev = g_ev[g_ev_next];
  g_ev_next = (g_ev_next + 1) % NEV;
  if (hipEventRecord(ev, reinterpret_cast<hipStream_t>(from)) != hipSuccess) return ST5_ERR_LAUNCH;
  if (hipStreamWaitEvent(reinterpret_cast<hipStream_t>(to), ev, 0) != hipSuccess) return ST5_ERR_LAUNCH;
  return ST5_OK;
}
