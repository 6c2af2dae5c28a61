// Library-level C ABI plumbing: version string and per-thread error message.
#include "common.h"
#include <string.h>

static thread_local char g_err[512] = "";

void gs_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* gs_version(void) { return "goslam_hip 0.1.0 (gfx950)"; }
extern "C" const char* gs_last_error(void) { return g_err; }

// ---------------------------------------------------------------- kernel timer -----------------------------------------
// bench.py's roofline entries need the duration of individual kernels measured live, on the stream they are launched on
// (torch.cuda.Event only sees torch's stream API; most ABI entries issue several kernels).  gs_timing_begin(stream)
// records a start event; from then on every kernel launch of this thread that goes through GS_CHECK_LAUNCH(name) is
// followed by an event on that stream, so the time between two consecutive events is the kernel launched between them
// (kernels of one stream execute in order; the events cost ~1 us of stream time each, which is why this is a
// measurement mode and not always on).  gs_timing_end() stops recording, gs_timing_read(name, ...) synchronises and sums.
#include <string>
#include <vector>

thread_local int gs_timing_on = 0;
namespace {
struct Mark { std::string name; hipEvent_t ev; };
thread_local std::vector<Mark> g_marks;
thread_local hipStream_t g_timing_stream = nullptr;
constexpr size_t MAX_MARKS = 1 << 16;

void drop_marks() {
  for (auto& m : g_marks) (void)hipEventDestroy(m.ev);
  g_marks.clear();
}
}  // namespace

void gs_timing_mark(const char* name) {
  if (!gs_timing_on || g_marks.size() >= MAX_MARKS) return;
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(g_timing_stream, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) return;
  hipEvent_t ev;
  if (hipEventCreate(&ev) != hipSuccess) return;
  if (hipEventRecord(ev, g_timing_stream) != hipSuccess) { (void)hipEventDestroy(ev); return; }
  g_marks.push_back({name, ev});
}

extern "C" int gs_timing_begin(gs_stream_t stream) {
  drop_marks();
  g_timing_stream = (hipStream_t)stream;
  gs_timing_on = 1;
  gs_timing_mark("");          // the start event
  if (g_marks.empty()) { gs_timing_on = 0; gs_set_error("timing_begin: cannot record an event on the stream"); return GS_ERR_LAUNCH; }
  return GS_OK;
}

extern "C" int gs_timing_end(void) {
  gs_timing_on = 0;
  return GS_OK;
}

extern "C" int gs_timing_read(const char* name, double* total_ms, int* count) {
  GS_REQUIRE(name && total_ms && count, "timing_read: null pointer");
  *total_ms = 0.0;
  *count = 0;
  for (size_t i = 1; i < g_marks.size(); ++i) {
    if (g_marks[i].name != name) continue;
    if (hipEventSynchronize(g_marks[i].ev) != hipSuccess) { gs_set_error("timing_read: event sync failed"); return GS_ERR_LAUNCH; }
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, g_marks[i - 1].ev, g_marks[i].ev) != hipSuccess) {
      gs_set_error("timing_read: elapsed time unavailable");
      return GS_ERR_LAUNCH;
    }
    *total_ms += ms;
    *count += 1;
  }
  return GS_OK;
}

extern "C" int gs_timing_names(char* buf, int buf_bytes) {
  GS_REQUIRE(buf && buf_bytes > 0, "timing_names: bad buffer");
  std::string out;
  std::vector<std::string> seen;
  for (size_t i = 1; i < g_marks.size(); ++i) {
    bool dup = false;
    for (auto& s : seen) dup = dup || s == g_marks[i].name;
    if (dup) continue;
    seen.push_back(g_marks[i].name);
    if (!out.empty()) out += ",";
    out += g_marks[i].name;
  }
  snprintf(buf, (size_t)buf_bytes, "%s", out.c_str());
  return GS_OK;
}
