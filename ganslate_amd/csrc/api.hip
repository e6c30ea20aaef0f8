// Library lifecycle + error plumbing for libganslate_hip.so (see include/ganslate_hip.h).
#include "common.hpp"
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <mutex>

namespace {
thread_local char g_err[512] = "";   // per host thread: autograd runs backward launches on its own threads, and a caller
                                      // reads the message right after the failing call on the thread that made it
void* g_zero = nullptr;      // 256-byte zero page: source of masked LDS-DMA lanes
void* g_dump = nullptr;      // 256 writable bytes nobody reads
// Loss reductions: 1024 partials + 1 arrival counter per workspace. Launches on one stream are ordered and share a
// workspace; launches on different streams may overlap (the discriminator pass runs beside the generators' backward),
// so every stream that ever launched a reduction owns one of GS_WS_SLOTS workspaces.
constexpr int GS_WS_SLOTS = 256;     // torch hands out streams from pools of 32 per priority: far below this
constexpr int GS_WS_FLOATS = 1040;
float* g_reduce_ws = nullptr;
void* g_ws_stream[GS_WS_SLOTS] = {};
int g_ws_used = 0;
std::mutex g_ws_mutex;
}  // namespace

void gs_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
const void* gs_zero_page() { return g_zero; }
void* gs_dump_page() { return g_dump; }
float* gs_reduce_workspace(void* stream) {
  if (!g_reduce_ws) return nullptr;
  std::lock_guard<std::mutex> lock(g_ws_mutex);
  for (int i = 0; i < g_ws_used; ++i)
    if (g_ws_stream[i] == stream) return g_reduce_ws + (size_t)i * GS_WS_FLOATS;
  if (g_ws_used < GS_WS_SLOTS) {
    g_ws_stream[g_ws_used] = stream;
    return g_reduce_ws + (size_t)(g_ws_used++) * GS_WS_FLOATS;
  }
  // more launching streams than workspaces: two live streams would share partial sums, so refuse instead of aliasing
  gs_set_error("loss reductions were launched from more than %d streams of this process; raise GS_WS_SLOTS (api.hip)",
               GS_WS_SLOTS);
  return nullptr;
}

namespace {
struct OptDef { const char* name; int value; };
OptDef g_opts[GS_OPT_COUNT] = {
    {"splitk", 1},              // split-K for launches with few output tiles and a long K loop (gconv.hip)
    {"splitk_max_blocks", 128}, // ... only below this many output tiles
    {"splitk_target", 256},     // ... aiming at this many workgroups
    {"hconv", 1},               // halo-resident forward kernel for narrow stride-1 layers (hconv.hip)
    {"hconv_wide", 1},          // halo-resident forward kernel for the wide 3x3 layers (hconvw.hip)
    {"hwgrad", 1},              // halo-resident weight-gradient kernels (hwgrad.hip)
    {"hwgrad_wide", 1},         // ... the wide 3x3 form
    {"hwgrad_planes", 1},       // ... 3x3x3 layers as three depth planes of it
    {"norm_bwd_ppb", 0},        // pixels per workgroup of the norm-backward reduction (0 = heuristic; tuning aid)
    {"norm_apply_unroll", 4},   // elements per thread of the norm-backward apply pass (tuning aid)
    {"gconv_tile288", 1},       // 288-pixel im2col tiles where they make exactly one round of workgroups (else 320)
    {"gconv_multi", 1},         // the parity classes of a stride-2 transposed conv / data gradient as one launch
    {"hconvw_ring", 1},         // fused data gradient of the reflect-padded wide 3x3 layers on the unpadded domain (hconvw.hip RING)
    {"hconvt", 192},            // halo-resident kernel for the four parity classes of a stride-2 layer in one pass (hconvt.hip):
                                // smallest grid (boxes x channel tiles x batch) it takes, 0 = off
    {"hstrip", 1024},           // halo-resident kernel for the W-folded k7 boundary convs (hstrip.hip): smallest grid, 0 = off
    {"wfold_rows", 1},          // row-staged forms of the four W-fold boundary transforms (wfold.hip) instead of one thread per pixel
    {"hwgrad_ft", 1},           // halo-resident weight gradient of narrow layers with few taps (hwgrad.hip: the 2-D k7 boundary convs)
    {"gconv_big", 192},         // smallest number of 256 x 128 im2col tiles that selects them (one workgroup per CU) over 128 x 128 (two)
    {"hconv_box8", 1},          // hconv.hip: 8 x 8 x 8 boxes on 8 waves for volumes (4 x 8 x 8 on 4 waves otherwise)
    {"hconvw_persist", 1},      // hconvw.hip: launches with more tiles than CUs run ceil(tiles / CUs) tiles per workgroup (0: one each)
    {"hstrip_regs", 1},         // hstrip.hip: persistent form with the weights in registers for the k7 boundary convs (0: one tile per workgroup)
    {"gconv_twin", 1},          // gconv.hip: twin batches on the im2col kernel as one launch (0: the two halves as two launches)
    {"wgrad_twin", 1},          // wgrad.hip: twin batches on the im2col weight-gradient kernel as one launch (0: two launches)
    {"gconv_smallk", 0},        // gconv.hip: layers with at most this many K-steps take 128 x 128 tiles on 8 waves (two workgroups per
                                // CU overlap each other's prologue / epilogue) instead of one 256 x 128 tile per CU; 0 = off
    {"gconv_persist", 16},      // pconv.hip: 256 x 128 im2col launches with more tiles than CUs and at most this many K-steps run as
                                // persistent workgroups (the K-step stream continues across tiles); 0 = off
    {"hconvt_persist", 1},      // hconvt.hip: launches with more tiles than CUs run as persistent workgroups (0: one tile each)
    {"ring_apply", 0},          // hconvw.hip: gs_gconv_ring_apply is offered (the consumer's norm backward inside the fused data gradient).
                                // OFF: the in-launch rendezvous costs more than the launch it saves (profiles/r05_ring_apply.txt);
                                // bits 2 / 4 / 8 / 16 are timing ablations (wrong results)
    {"norm_xcd", 0},            // norm.hip: the channel-group norm kernels take image n on XCD n % 8 (1: last image first, 2: in order)
    {"wgrad_rows", 1},          // wgrad.hip: the im2col weight gradient stores whole tile rows through LDS; one split adds without atomics
    {"splitk_multi", 1},        // gconv.hip: split-K over the merged parity classes of a small stride-2 layer (one launch + one finalize)
    {"splitk_ring", 1},         // gconv.hip: split-K launches of the 128 x 128 tile run a 4-stage ring (three K-steps of cold weights in flight)
    {"gconv_ring4", 16},        // gconv.hip: 128-pixel im2col tiles in a grid of <= 2 workgroups per CU with at least this many K-steps run a 4-stage ring; 0 = off
    {"ring_dbg", 0},            // hconvw.hip RING: timing ablations (wrong results): 1 no ring MFMAs, 2 no y / g2 loads, 4 no sums, 8 no ring adds
    {"hconv5", 64},             // hconv5.hip: register-resident-weights kernel for the 16 -> 16 channel k5 volume convs; smallest volume
                                // (batch x voxels / 2048) it takes (0 = off)
    {"hconv5_seg", 0},          // ... z segments per column (0 = as many as fill the chip; tests force long segments with 1 / 2)
    {"hwgrad2", 2},             // hwgrad.hip: double-buffered, decode-once form of the narrow volume weight gradient with 65..128 taps
                                // (>= 2: also for layers wide on both sides, 33..64 x 17..64 channels, instead of the im2col kernel)
    {"hconv2", 4},              // hconv.hip: persistent double-buffered form of the narrow volume forward / data-gradient kernel (17..64 output channels;
                                // >= 2: also 64 -> 64 channels — wide on both sides — instead of the split-K im2col launch + its finalize;
                                // >= 3: 32-channel layers on <= 128 boxes as two 16-channel groups per box; >= 4: 64-channel layers on <= 64 boxes as four)
    {"pwise", 8},               // pwise.hip: register-operand kernels for one-tap layers with <= 8 channels on one side (smallest volume in 2048-voxel units, 0 = off)
    {"adam_blocks", 8192},      // optim.hip: largest grid of the Adam update (the chunks launched under a backward pass take fewer: they
                                // must not crowd the pass's own launches out of the CUs)
};
}  // namespace
int gs_opt(int id) { return g_opts[id].value; }

extern "C" int gs_set_option(const char* name, int value) {
  for (int i = 0; i < GS_OPT_COUNT; ++i)
    if (name && strcmp(name, g_opts[i].name) == 0) { g_opts[i].value = value; return 0; }
  gs_set_error("gs_set_option: unknown option '%s'", name ? name : "(null)");
  return 2;
}
extern "C" int gs_get_option(const char* name, int* value) {
  for (int i = 0; i < GS_OPT_COUNT; ++i)
    if (name && value && strcmp(name, g_opts[i].name) == 0) { *value = g_opts[i].value; return 0; }
  gs_set_error("gs_get_option: unknown option '%s'", name ? name : "(null)");
  return 2;
}

extern "C" const char* gs_last_error(void) { return g_err; }

extern "C" int gs_init(int device) {
  GS_CHECK_HIP(hipSetDevice(device));
  if (!g_zero) {
    GS_CHECK_HIP(hipMalloc(&g_zero, 256));
    GS_CHECK_HIP(hipMemset(g_zero, 0, 256));
  }
  if (!g_dump) GS_CHECK_HIP(hipMalloc(&g_dump, 256));
  if (!g_reduce_ws) {
    GS_CHECK_HIP(hipMalloc(reinterpret_cast<void**>(&g_reduce_ws), GS_WS_SLOTS * GS_WS_FLOATS * sizeof(float)));
    GS_CHECK_HIP(hipMemset(g_reduce_ws, 0, GS_WS_SLOTS * GS_WS_FLOATS * sizeof(float)));
  }
  hipDeviceProp_t prop;
  GS_CHECK_HIP(hipGetDeviceProperties(&prop, device));
  GS_REQUIRE(strncmp(prop.gcnArchName, "gfx950", 6) == 0,
             "gs_init: device %d is %s; this library is built for gfx950 (MI355X) only", device, prop.gcnArchName);
  return 0;
}

extern "C" void gs_shutdown(void) {
  if (g_zero) { (void)hipFree(g_zero); g_zero = nullptr; }
  if (g_dump) { (void)hipFree(g_dump); g_dump = nullptr; }
  if (g_reduce_ws) { (void)hipFree(g_reduce_ws); g_reduce_ws = nullptr; g_ws_used = 0; }
}
