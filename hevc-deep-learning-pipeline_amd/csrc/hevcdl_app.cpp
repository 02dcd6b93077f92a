// hevcdl_app.cpp -- command-line front end with the reference application's surface.
//
// Accepts what the reference is started with,  TAppEncoder -c encoder_intra_main.cfg -c bitstream.cfg [-q QP ...]
//   option syntax                HM_dl/source/Lib/TAppCommon/program_options_lite.cpp (cfg "Key : value  # comment",
//                                "--Key=value" / "--Key value", short options)
//   option names / short forms   HM_dl/source/App/TAppEncoder/TAppEncCfg.cpp:730-1290
//   planar YUV reader            HM_dl/source/Lib/TLibVideoIO/TVideoIOYuv.cpp:249,675 (8-bit 4:2:0 file, FrameSkip)
//   picture log line / summary   TEncGOP.cpp:2500-2541, TEncAnalyze.h:163-370
// and drives the GPU path through the C ABI of include/hevcdl.h only.  Keys that would change the path are checked
// against what the path implements (rejected, not ignored).  Labels come from the on-device CNN, or -- the reference's own
// file IPC format -- from --LabelDir <dir>/<frame>/ctu<addr>.txt (16 integers, TEncCu.cpp:255-262).
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <memory>
#include <mutex>
#include <functional>
#include <dlfcn.h>
#include <sys/types.h>
#include <sstream>
#include <string>
#include <atomic>
#include <thread>
#include <vector>
#include "hevcdl.h"

namespace {

enum Kind { USED, PATH, STAGE, NOEFFECT };
struct Key { const char *name; const char *shortopt; Kind kind; const char *required; };   // required: value the path implements (PATH)

const Key KEYS[] = {
  // used by this front end
  { "InputFile", "i", USED, 0 }, { "BitstreamFile", "b", USED, 0 }, { "ReconFile", "o", USED, 0 }, { "SourceWidth", "wdt", USED, 0 },
  { "SourceHeight", "hgt", USED, 0 }, { "FrameRate", "fr", USED, 0 }, { "FrameSkip", "fs", USED, 0 }, { "FramesToBeEncoded", "f", USED, 0 },
  { "QP", "q", USED, 0 },
  // extensions of this front end
  { "LabelDir", 0, USED, 0 }, { "BatchFrames", 0, USED, 0 }, { "ChunkFrames", 0, USED, 0 }, { "Device", 0, USED, 0 }, { "Devices", 0, USED, 0 }, { "NumDevices", 0, USED, 0 }, { "Weights", 0, USED, 0 }, { "RecordFile", 0, USED, 0 },
  { "CnnInput", 0, USED, 0 }, { "BnMode", 0, USED, 0 }, { "PrintConfig", 0, USED, 0 }, { "LoopFilterDisable", 0, USED, 0 },
  // keys that define the path: only the implemented value is accepted
  { "InputBitDepth", 0, USED, 0 }, { "InternalBitDepth", 0, USED, 0 }, { "InputChromaFormat", 0, PATH, "420" }, { "Profile", 0, USED, 0 },
  { "MaxCUWidth", 0, PATH, "64" }, { "MaxCUHeight", 0, PATH, "64" }, { "MaxPartitionDepth", 0, PATH, "4" },
  { "QuadtreeTULog2MaxSize", 0, PATH, "5" }, { "QuadtreeTULog2MinSize", 0, PATH, "2" }, { "QuadtreeTUMaxDepthIntra", 0, PATH, "3" },
  { "IntraPeriod", 0, PATH, "1" }, { "GOPSize", 0, PATH, "1" }, { "MaxDeltaQP", 0, PATH, "0" }, { "DeltaQpRD", 0, PATH, "0" },
  { "RDOQ", 0, USED, 0 }, { "RDOQTS", 0, USED, 0 }, { "TransformSkip", 0, USED, 0 }, { "TransformSkipFast", 0, USED, 0 },
  { "SignHideFlag", "SBH", USED, 0 }, { "StrongIntraSmoothing", 0, USED, 0 }, { "FastUDIUseMPMEnabled", 0, USED, 0 },      // tool switches (hevcdl_config.tools): 0 or 1
  { "SliceMode", 0, PATH, "0" }, { "PCMEnabledFlag", 0, PATH, "0" }, { "NumTileColumnsMinus1", 0, USED, 0 },
  { "NumTileRowsMinus1", 0, USED, 0 }, { "WaveFrontSynchro", 0, USED, 0 }, { "ScalingList", 0, PATH, "0" },
  { "TransquantBypassEnable", 0, PATH, "0" }, { "CUTransquantBypassFlagForce", 0, PATH, "0" },
  // stream / in-loop filter keys
  { "Level", 0, USED, 0 }, { "DecodingRefreshType", 0, PATH, "1" }, { "ReWriteParamSetsFlag", 0, USED, 0 }, { "LoopFilterOffsetInPPS", 0, PATH, "1" },
  { "LoopFilterBetaOffset_div2", 0, USED, 0 }, { "LoopFilterTcOffset_div2", 0, USED, 0 },
  { "DeblockingFilterMetric", 0, PATH, "0" }, { "SAO", 0, USED, 0 }, { "SAOLcuBoundary", 0, PATH, "0" }, { "LFCrossSliceBoundaryFlag", 0, NOEFFECT, 0 },      // (without slices the reference sets it to 1 whatever the cfg says: TAppEncTop.cpp:278-281)
  { "LFCrossTileBoundaryFlag", 0, USED, 0 }, { "SEIDecodedPictureHash", 0, USED, 0 },
  // no effect on an all-intra slice with the settings above
  { "QuadtreeTUMaxDepthInter", 0, NOEFFECT, 0 }, { "FastSearch", 0, NOEFFECT, 0 }, { "SearchRange", 0, NOEFFECT, 0 }, { "HadamardME", 0, NOEFFECT, 0 },
  { "FEN", 0, NOEFFECT, 0 }, { "FDM", 0, NOEFFECT, 0 }, { "AMP", 0, NOEFFECT, 0 }, { "MaxCuDQPDepth", 0, NOEFFECT, 0 }, { "SliceArgument", 0, NOEFFECT, 0 },
  { "PCMLog2MaxSize", 0, NOEFFECT, 0 }, { "PCMLog2MinSize", 0, NOEFFECT, 0 }, { "PCMInputBitDepthFlag", 0, NOEFFECT, 0 },
  { "PCMFilterDisableFlag", 0, NOEFFECT, 0 }, { "TileUniformSpacing", 0, USED, 0 }, { "TileColumnWidthArray", 0, USED, 0 },
  { "TileRowHeightArray", 0, USED, 0 }, { "ScalingListFile", 0, NOEFFECT, 0 },
};

const Key *find_key(const std::string &name, bool shortform)
{
  for (const Key &k : KEYS) if (shortform ? (k.shortopt && name == k.shortopt) : name == k.name) return &k;
  return nullptr;
}

std::string trim(const std::string &s)
{
  size_t a = s.find_first_not_of(" \t\r\n"), b = s.find_last_not_of(" \t\r\n");
  return a == std::string::npos ? "" : s.substr(a, b - a + 1);
}

struct Options {
  std::map<std::string, std::string> v;
  std::vector<std::string> errors;
  void set(const std::string &name, const std::string &val, bool shortform, const std::string &where)
  {
    const Key *k = find_key(name, shortform);
    if (!k) { errors.push_back(where + ": unknown option '" + name + "'"); return; }
    v[k->name] = val;
  }
  bool parse_cfg(const std::string &path)
  { // program_options_lite::scanFile: "name : value", '#' starts a comment, blank lines ignored
    std::ifstream f(path);
    if (!f) { errors.push_back("cannot open configuration file '" + path + "'"); return false; }
    std::string line; int ln = 0;
    while (std::getline(f, line)) {
      ln++;
      const size_t h = line.find('#'); if (h != std::string::npos) line.erase(h);
      line = trim(line); if (line.empty()) continue;
      const size_t c = line.find(':');
      if (c == std::string::npos) { errors.push_back(path + ":" + std::to_string(ln) + ": expected 'name : value'"); continue; }
      set(trim(line.substr(0, c)), trim(line.substr(c + 1)), false, path + ":" + std::to_string(ln));
    }
    return true;
  }
  void parse_argv(int argc, char **argv)
  {
    for (int i = 1; i < argc; i++) {
      std::string a = argv[i];
      if (a == "-c") { if (i + 1 < argc) parse_cfg(argv[++i]); else errors.push_back("-c needs a file"); continue; }
      if (a.rfind("--", 0) == 0) {
        const size_t e = a.find('=');
        if (e != std::string::npos) set(a.substr(2, e - 2), a.substr(e + 1), false, "command line");
        else if (a.substr(2) == "PrintConfig") set("PrintConfig", "1", false, "command line");
        else if (i + 1 < argc) { set(a.substr(2), argv[i + 1], false, "command line"); i++; }
        else errors.push_back("option '" + a + "' needs a value");
      } else if (a.size() > 1 && a[0] == '-') {
        if (i + 1 < argc) { set(a.substr(1), argv[i + 1], true, "command line"); i++; }
        else errors.push_back("option '" + a + "' needs a value");
      } else errors.push_back("stray argument '" + a + "'");
    }
  }
  std::string get(const char *k, const char *dflt = "") const { auto it = v.find(k); return it == v.end() ? dflt : it->second; }
  long geti(const char *k, long dflt) const { auto it = v.find(k); return it == v.end() ? dflt : strtol(it->second.c_str(), nullptr, 10); }
};

std::string native_path(std::string p)
{ // the reference's cfg files carry Windows paths (".\rec\rec.yuv")
  std::replace(p.begin(), p.end(), '\\', '/');
  return p;
}

double psnr_of(unsigned long long sse, double n, double maxval = 255.0) { return sse == 0 ? 999.99 : 10.0 * log10(maxval * maxval * n / (double)sse); }   // TEncGOP.cpp:2391-2393, maxval = 255 << (bitDepth - 8)

std::string json_escape(const std::string &s) { std::string o; for (char c : s) { if (c == '"' || c == '\\') o += '\\'; o += c; } return o; }

// ---- RCCL from C++: the per-picture rows of every device in one all-gather -------------------------------------------------------------------
// librccl and the HIP runtime are opened at run time (a single-device run needs neither); the handful of entry points used are declared here with their ABI types
// (rccl.h: ncclResult_t / ncclDataType_t are ints, ncclComm_t and hipStream_t opaque pointers; ncclUint64 = 5).
struct RcclApi {
  void *h_rccl = nullptr, *h_hip = nullptr;
  int (*CommInitAll)(void **comms, int ndev, const int *devlist) = nullptr;
  int (*CommDestroy)(void *comm) = nullptr;
  int (*AllGather)(const void *send, void *recv, size_t count, int datatype, void *comm, void *stream) = nullptr;
  int (*GroupStart)() = nullptr; int (*GroupEnd)() = nullptr;
  const char *(*GetErrorString)(int) = nullptr;
  int (*GetVersion)(int *) = nullptr; int (*CommCount)(void *comm, int *count) = nullptr;
  int (*hipSetDevice)(int) = nullptr; int (*hipMalloc)(void **, size_t) = nullptr; int (*hipFree)(void *) = nullptr;
  int (*hipMemcpy)(void *, const void *, size_t, int) = nullptr; int (*hipDeviceSynchronize)() = nullptr;
  bool load(std::string &err)
  {
    h_rccl = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL); if (!h_rccl) h_rccl = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    h_hip = dlopen("libamdhip64.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h_rccl || !h_hip) { err = std::string("cannot load ") + (h_rccl ? "libamdhip64.so" : "librccl.so") + ": " + dlerror(); return false; }
#define SYM(lib, field, name) do { *(void **)&field = dlsym(lib, name); if (!field) { err = std::string("missing symbol ") + name; return false; } } while (0)
    SYM(h_rccl, CommInitAll, "ncclCommInitAll"); SYM(h_rccl, CommDestroy, "ncclCommDestroy"); SYM(h_rccl, AllGather, "ncclAllGather");
    SYM(h_rccl, GroupStart, "ncclGroupStart"); SYM(h_rccl, GroupEnd, "ncclGroupEnd"); SYM(h_rccl, GetErrorString, "ncclGetErrorString");
    SYM(h_rccl, GetVersion, "ncclGetVersion"); SYM(h_rccl, CommCount, "ncclCommCount");
    SYM(h_hip, hipSetDevice, "hipSetDevice"); SYM(h_hip, hipMalloc, "hipMalloc"); SYM(h_hip, hipFree, "hipFree"); SYM(h_hip, hipMemcpy, "hipMemcpy");
    SYM(h_hip, hipDeviceSynchronize, "hipDeviceSynchronize");
#undef SYM
    return true;
  }
};

// rows_of(i): the rows (row_words 64-bit words each) of block i, coded on device dev_of(i).  -> table: every rank's contribution, padded with rows whose first word is ~0
// info: what ran, for the log -- the library's version, the size the communicator reports (ncclCommCount), the devices
bool gather_rows_rccl(size_t n_blocks, const std::function<int(size_t)> &dev_of, const std::function<const std::vector<unsigned long long> &(size_t)> &rows_of, int row_words,
                      std::vector<unsigned long long> &table, std::string &err, std::string *info = nullptr)
{
  RcclApi api;
  if (!api.load(err)) return false;
  std::vector<int> ranks;                                  // physical devices, in order of first appearance: one RCCL rank each
  for (size_t i = 0; i < n_blocks; i++) if (std::find(ranks.begin(), ranks.end(), dev_of(i)) == ranks.end()) ranks.push_back(dev_of(i));
  const int R = (int)ranks.size();
  std::vector<std::vector<unsigned long long>> contrib(R);
  for (size_t i = 0; i < n_blocks; i++) { auto &c = contrib[std::find(ranks.begin(), ranks.end(), dev_of(i)) - ranks.begin()]; const auto &r = rows_of(i); c.insert(c.end(), r.begin(), r.end()); }
  size_t words = (size_t)row_words;
  for (const auto &c : contrib) words = std::max(words, c.size());
  for (auto &c : contrib) c.resize(words, ~0ull);
  std::vector<void *> comms(R, nullptr), d_send(R, nullptr), d_recv(R, nullptr);
  auto cleanup = [&]() { for (int r = 0; r < R; r++) { api.hipSetDevice(ranks[r]); if (d_send[r]) api.hipFree(d_send[r]); if (d_recv[r]) api.hipFree(d_recv[r]); if (comms[r]) api.CommDestroy(comms[r]); } };
  auto fail = [&](const std::string &what, int code, bool nccl) { err = what + (nccl ? std::string(": ") + api.GetErrorString(code) : " failed (hip error " + std::to_string(code) + ")"); cleanup(); return false; };
  int e = api.CommInitAll(comms.data(), R, ranks.data());
  if (e) { std::fill(comms.begin(), comms.end(), nullptr); return fail("ncclCommInitAll", e, true); }
  for (int r = 0; r < R; r++) {
    if ((e = api.hipSetDevice(ranks[r]))) return fail("hipSetDevice", e, false);
    if ((e = api.hipMalloc(&d_send[r], words * 8)) || (e = api.hipMalloc(&d_recv[r], words * 8 * (size_t)R))) return fail("hipMalloc", e, false);
    if ((e = api.hipMemcpy(d_send[r], contrib[r].data(), words * 8, 1 /* hipMemcpyHostToDevice */))) return fail("hipMemcpy", e, false);
  }
  if ((e = api.GroupStart())) return fail("ncclGroupStart", e, true);
  for (int r = 0; r < R; r++) { api.hipSetDevice(ranks[r]); if ((e = api.AllGather(d_send[r], d_recv[r], words, 5 /* ncclUint64 */, comms[r], nullptr))) { api.GroupEnd(); return fail("ncclAllGather", e, true); } }
  if ((e = api.GroupEnd())) return fail("ncclGroupEnd", e, true);
  for (int r = 0; r < R; r++) { api.hipSetDevice(ranks[r]); if ((e = api.hipDeviceSynchronize())) return fail("hipDeviceSynchronize", e, false); }
  table.resize(words * (size_t)R);
  api.hipSetDevice(ranks[0]);
  if ((e = api.hipMemcpy(table.data(), d_recv[0], words * 8 * (size_t)R, 2 /* hipMemcpyDeviceToHost */))) return fail("hipMemcpy", e, false);
  if (info) {
    int ver = 0, cnt = 0; api.GetVersion(&ver); api.CommCount(comms[0], &cnt);
    char b[256]; int o = snprintf(b, sizeof b, "ncclAllGather over %d rank%s (ncclCommCount %d, RCCL/NCCL version code %d, one rank per physical device:", R, R == 1 ? "" : "s", cnt, ver);
    for (int r = 0; r < R && o < (int)sizeof b - 8; r++) o += snprintf(b + o, sizeof b - o, " %d", ranks[r]);
    snprintf(b + o, sizeof b - o, "), %zu words each", words);
    *info = b;
  }
  cleanup();
  return true;
}

} // namespace

int main(int argc, char **argv)
{
  const std::chrono::steady_clock::time_point t_main0 = std::chrono::steady_clock::now();
  Options opt;
  opt.parse_argv(argc, argv);
  // keys that define the path must carry the implemented value
  std::vector<std::string> stage_keys;
  for (const auto &kv : opt.v) {
    const Key *k = find_key(kv.first, false);
    if (k->kind == PATH && kv.second != k->required)
      opt.errors.push_back(std::string(k->name) + " = " + kv.second + " is not implemented by this path (only " + k->required + ")");
    if (k->kind == STAGE) stage_keys.push_back(k->name);
  }
  const int width = (int)opt.geti("SourceWidth", 0), height = (int)opt.geti("SourceHeight", 0), qp = (int)opt.geti("QP", 30);
  const long frame_skip = opt.geti("FrameSkip", 0); long n_frames = opt.geti("FramesToBeEncoded", 0);
  const double fps = atof(opt.get("FrameRate", "30").c_str());
  const std::string input = native_path(opt.get("InputFile")), recon_path = native_path(opt.get("ReconFile")), label_dir = native_path(opt.get("LabelDir"));
  const std::string cnn_input = opt.get("CnnInput", "rgb601");
  const std::string bitstream_path = native_path(opt.get("BitstreamFile"));
  const int level_idc = (int)(atof(opt.get("Level", "6.2").c_str()) * 30.0 + 0.5);      // general_level_idc
  const bool deblock = opt.geti("LoopFilterDisable", 0) == 0;
  const bool sao = opt.geti("SAO", 1) != 0;                                    // TAppEncCfg.cpp: SAO defaults to on
  // sample bit depth (TAppEncCfg.cpp:770-780): 8 (Profile main) or 10 (Profile main10; the file then holds 16-bit little-endian samples).
  // InternalBitDepth 0 = the input's; bit-depth conversion between file and codec is not implemented.
  const int in_bd = (int)opt.geti("InputBitDepth", 8), bit_depth = opt.geti("InternalBitDepth", 0) == 0 ? in_bd : (int)opt.geti("InternalBitDepth", 0);
  if (bit_depth != in_bd) opt.errors.push_back("InternalBitDepth must equal InputBitDepth on this path (no bit-depth conversion)");
  if (bit_depth != 8 && bit_depth != 10) opt.errors.push_back("InternalBitDepth = " + std::to_string(bit_depth) + " is not implemented by this path (only 8 and 10)");
  { const std::string prof = opt.get("Profile", bit_depth == 8 ? "main" : "main10");
    if (prof != (bit_depth == 8 ? "main" : "main10")) opt.errors.push_back("Profile = " + prof + " is not implemented by this path (main at 8 bits, main10 at 10 bits)"); }
  // decoded picture hash SEI (TAppEncCfg.cpp:1093): 0 none, 1 MD5 of the output picture behind every access unit
  const int hash_sei = (int)opt.geti("SEIDecodedPictureHash", 0);
  for (const char *key : { "LoopFilterBetaOffset_div2", "LoopFilterTcOffset_div2" }) {      // the reference's own range check (TAppEncCfg.cpp xConfirmPara: -6 .. 6)
    const long v = opt.geti(key, 0);
    if (v < -6 || v > 6) opt.errors.push_back(std::string(key) + " = " + std::to_string(v) + " is out of range (-6 .. 6)");
  }
  if (hash_sei < 0 || hash_sei > 3) opt.errors.push_back("SEIDecodedPictureHash = " + std::to_string(hash_sei) + " is not a hash of the reference (0 none, 1 MD5, 2 CRC, 3 checksum)");
  // tiles (TAppEncCfg.cpp:1024-1028): uniformly spaced columns x rows or explicit sizes; LFCrossTileBoundaryFlag (default 1) lets the in-loop filters cross tile borders
  const int tile_cols = (int)opt.geti("NumTileColumnsMinus1", 0) + 1, tile_rows = (int)opt.geti("NumTileRowsMinus1", 0) + 1;
  const int tile_uniform = (int)opt.geti("TileUniformSpacing", 0) != 0;
  std::vector<int> tile_cw, tile_rh;               // TileColumnWidthArray / TileRowHeightArray: sizes in CTUs of all but the last column / row
  auto parse_ints = [](const std::string &text, std::vector<int> &out) { std::string t = text; for (char &ch : t) if (ch == ',') ch = ' '; std::istringstream is(t); int v; while (is >> v) out.push_back(v); };
  // WaveFrontSynchro (TAppEncCfg.cpp:975): 1 = entropy_coding_sync_enabled_flag; CTU rows start from the contexts behind the second CTU of the row above and are walked by
  // waves of their own on the device.  The reference refuses the key together with tiles outside the high-throughput profile (TAppEncCfg.cpp xCheckParameter); so does this path.
  const long wavefront = opt.geti("WaveFrontSynchro", 0);
  if (wavefront != 0 && wavefront != 1) opt.errors.push_back("WaveFrontSynchro = " + std::to_string(wavefront) + " is not a value of the key (0 or 1)");
  if (wavefront && tile_cols * tile_rows > 1) opt.errors.push_back("Tiles and entropy-coding-sync (Wavefronts) can not be applied together (as the reference, outside its high-throughput profile)");
  if (tile_cols * tile_rows > 1) {
    if (!tile_uniform) {
      parse_ints(opt.get("TileColumnWidthArray"), tile_cw); parse_ints(opt.get("TileRowHeightArray"), tile_rh);
      if ((int)tile_cw.size() < tile_cols - 1 || (int)tile_rh.size() < tile_rows - 1 || tile_cols > 20 || tile_rows > 22)
        opt.errors.push_back("TileUniformSpacing = 0 needs TileColumnWidthArray / TileRowHeightArray with a size for every tile column / row but the last");
    }
  }
  if (opt.v.count("PrintConfig")) {
    printf("{\"InputFile\": \"%s\", \"ReconFile\": \"%s\", \"SourceWidth\": %d, \"SourceHeight\": %d, \"QP\": %d, \"FrameSkip\": %ld, \"FramesToBeEncoded\": %ld, "
           "\"FrameRate\": %g, \"LabelDir\": \"%s\", \"CnnInput\": \"%s\", \"BitstreamFile\": \"%s\", \"level_idc\": %d, \"tiles\": [%d, %d], \"bit_depth\": %d, \"stage_keys\": [", json_escape(input).c_str(), json_escape(recon_path).c_str(), width, height, qp,
           frame_skip, n_frames, fps, json_escape(label_dir).c_str(), cnn_input.c_str(), json_escape(bitstream_path).c_str(), level_idc, tile_cols, tile_rows, bit_depth);
    for (size_t i = 0; i < stage_keys.size(); i++) printf("%s\"%s\"", i ? ", " : "", stage_keys[i].c_str());
    printf("], \"errors\": [");
    for (size_t i = 0; i < opt.errors.size(); i++) printf("%s\"%s\"", i ? ", " : "", json_escape(opt.errors[i]).c_str());
    printf("]}\n");
    return opt.errors.empty() ? 0 : 2;
  }
  if (input.empty()) opt.errors.push_back("InputFile (-i) is required");
  if (width <= 0 || height <= 0) opt.errors.push_back("SourceWidth / SourceHeight (-wdt / -hgt) are required");
  if (cnn_input != "rgb601" && cnn_input != "luma") opt.errors.push_back("CnnInput must be rgb601 or luma");
  const std::string bn_mode = opt.get("BnMode", "reference");        // reference: training-mode BatchNorm as use_model.py runs it; eval: running statistics
  if (bn_mode != "reference" && bn_mode != "eval") opt.errors.push_back("BnMode must be reference or eval");
  if (!opt.errors.empty()) { for (const auto &e : opt.errors) fprintf(stderr, "Error: %s\n", e.c_str()); return 2; }

  const size_t frame_bytes = hevcdl_frame_bytes_bd(width, height, bit_depth);
  FILE *fin = fopen(input.c_str(), "rb");
  if (!fin) { fprintf(stderr, "Error: cannot open input file '%s'\n", input.c_str()); return 2; }
  fseek(fin, 0, SEEK_END); const long long fsize = ftell(fin);
  fclose(fin);
  const long avail = (long)(fsize / (long long)frame_bytes) - frame_skip;
  if (avail <= 0) { fprintf(stderr, "Error: input holds no frame after FrameSkip\n"); return 2; }
  if (n_frames <= 0 || n_frames > avail) n_frames = avail;                    // TAppEncTop: stops at end of file
  // Devices.  --Device d (default 0): one device.  --Devices 0-7 | 0,2,5 | --NumDevices N: the frames of the job in contiguous blocks, one block, one host thread and
  // one context per entry (the app loop of TAppEncTop.cpp:568-691 once per block; -fs / -f of TAppEncCfg.cpp:786,788 are what a block is); an entry may repeat a
  // device (two contexts share it: the contexts then never wait on each other, HEVCDL_EXEC_NO_UNIT_HANDOVER).
  std::vector<int> devices;
  if (opt.v.count("Devices")) {
    std::string t = opt.get("Devices"); for (char &ch : t) if (ch == ',') ch = ' ';
    std::istringstream is(t); std::string tok;
    while (is >> tok) { const size_t d = tok.find('-'); if (d != std::string::npos && d > 0) { for (int v = atoi(tok.substr(0, d).c_str()); v <= atoi(tok.substr(d + 1).c_str()); v++) devices.push_back(v); } else devices.push_back(atoi(tok.c_str())); }
  } else if (opt.v.count("NumDevices")) { for (int v = 0; v < (int)opt.geti("NumDevices", 1); v++) devices.push_back(v); }
  else devices.push_back((int)opt.geti("Device", 0));
  if (devices.empty() || devices.size() > 64) { fprintf(stderr, "Error: Devices / NumDevices must name 1..64 devices\n"); return 2; }
  const int n_shards = (int)std::min<long>((long)devices.size(), n_frames);
  const bool multi = devices.size() > 1;
  bool shared_device = false;
  for (size_t i = 0; i < devices.size(); i++) for (size_t j = 0; j < i; j++) if (devices[i] == devices[j]) shared_device = true;
  const long per_shard = (n_frames + n_shards - 1) / n_shards;                // the largest block: sharding.shard_frames deals contiguous blocks [i * n / shards, (i + 1) * n / shards), none empty
  // Pictures per device call.  A frame is a serial chain of CTUs: a call costs about the same from 1 to ~250 pictures and grows slowly beyond (the decision kernel
  // keeps one workgroup per CU busy with 1..8 pictures), so the default is a device's whole share, bounded by the 2048 (picture, tile) units the GPU holds, by 24 GiB
  // of originals on the host and by the DEVICE's free memory: a picture of a call occupies 3 frame buffers (original, reconstruction, filtered picture), its CTU
  // records, labels / logits and SAO parameters in HBM (hevcdl_create is retried with half the batch when the device refuses).  The results of a call come back chunk by
  // chunk (hevcdl_encode_pictures_chunked): the host never holds more than the originals of a batch and two chunks of results.  Several device calls: equal shares.
  const int ctus = hevcdl_ctus_per_frame(width, height);
  const long per_picture_dev = 3 * (long)frame_bytes + (long)ctus * ((long)sizeof(hevcdl_ctu_record) + 16 + 256 + 3 * 140) + (2L << 20);
  long dev_cap = 2048;
  { size_t free_b = 0, total_b = 0;       // the smallest free memory over the devices named: every shard's batch is sized from it
    for (size_t i = 0; i < devices.size(); i++) { size_t f = 0, t = 0; if (hevcdl_device_memory(devices[i], &f, &t) == HEVCDL_OK && f > 0 && (free_b == 0 || f < free_b)) { free_b = f; total_b = t; } }
    const size_t workspace = (size_t)4400 << 20;        // the decision kernel's workspace: up to 1.6 MB x 10 waves x 256 workgroups per context (hevcdl_reserve_workspace)
    if (free_b > 0) { double mine = (double)free_b * 0.8 / (shared_device ? (double)devices.size() : 1.0);       // (contexts that share a device share its memory)
      if (mine > 2.0 * (double)workspace) mine -= (double)workspace;
      dev_cap = std::max<long>(1, (long)mine / per_picture_dev); } }
  const long auto_batch = std::max<long>(1, std::min<long>(std::min<long>(2048 / (tile_cols * tile_rows), dev_cap), (24L << 30) / (long)frame_bytes));
  const long even_batch = (per_shard + ((per_shard + auto_batch - 1) / auto_batch) - 1) / ((per_shard + auto_batch - 1) / auto_batch);
  int batch = (int)std::min<long>(per_shard, std::max<long>(1, opt.geti("BatchFrames", even_batch)));

  hevcdl_config cfg;
  hevcdl_status st = hevcdl_config_default_bd(&cfg, width, height, qp, bit_depth);
  if (st != HEVCDL_OK) { fprintf(stderr, "Error: unsupported picture size / QP (status %d)\n", (int)st); return 2; }
  cfg.tile_columns = tile_cols; cfg.tile_rows = tile_rows; cfg.wavefront = (int)wavefront;
  if (tile_cols * tile_rows > 1) {
    cfg.tile_uniform_spacing = tile_uniform; cfg.lf_across_tiles = opt.geti("LFCrossTileBoundaryFlag", 1) != 0;
    if (!tile_uniform) { for (int i = 0; i < tile_cols - 1; i++) cfg.tile_column_width[i] = tile_cw[i]; for (int i = 0; i < tile_rows - 1; i++) cfg.tile_row_height[i] = tile_rh[i]; }
  }
  // tool switches of the cfg (TAppEncCfg.cpp:900-901,917-918,950,978,1007; encoder_intra_main.cfg:37-38,53-54): defaults as the reference's
  uint32_t tools = HEVCDL_TOOLS_REFERENCE;
  if (opt.geti("RDOQ", 1) == 0) tools &= ~HEVCDL_TOOL_RDOQ;
  if (opt.geti("RDOQTS", 1) == 0) tools &= ~HEVCDL_TOOL_RDOQTS;
  if (opt.geti("TransformSkip", 1) == 0) tools &= ~HEVCDL_TOOL_TSKIP;
  if (opt.geti("TransformSkipFast", 1) == 0) tools &= ~HEVCDL_TOOL_TSKIP_FAST;
  if (opt.geti("SignHideFlag", 1) == 0) tools &= ~HEVCDL_TOOL_SIGN_HIDE;
  if (opt.geti("StrongIntraSmoothing", 1) == 0) tools &= ~HEVCDL_TOOL_STRONG_INTRA;
  if (opt.geti("FastUDIUseMPMEnabled", 1) == 0) tools &= ~HEVCDL_TOOL_FAST_UDI_MPM;
  cfg.tools = tools;
  cfg.lf_beta_offset_div2 = (int)opt.geti("LoopFilterBetaOffset_div2", 0); cfg.lf_tc_offset_div2 = (int)opt.geti("LoopFilterTcOffset_div2", 0);      // -6 .. 6 (hevcdl_create checks)
  cfg.cnn_input = cnn_input == "luma" ? HEVCDL_CNN_INPUT_LUMA : HEVCDL_CNN_INPUT_RGB601;
  cfg.bn_mode = bn_mode == "eval" ? HEVCDL_BN_EVAL : HEVCDL_BN_REFERENCE;
  if (shared_device) cfg.exec_flags |= HEVCDL_EXEC_NO_UNIT_HANDOVER;
  std::string wpath = opt.get("Weights");
  if (wpath.empty()) { // next to the library: <pkg>/weights/hevc_encoder_model.f32, this binary lives in <pkg>/bin
    std::string self = argv[0]; const size_t s1 = self.find_last_of('/'); self = s1 == std::string::npos ? "." : self.substr(0, s1);
    wpath = self + "/../weights/hevc_encoder_model.f32";
  }
  std::vector<float> weights(HEVCDL_WEIGHT_FLOATS);
  { FILE *fw = fopen(wpath.c_str(), "rb");
    if (!fw || fread(weights.data(), sizeof(float), weights.size(), fw) != weights.size()) { fprintf(stderr, "Error: cannot read %d weights from '%s'\n", (int)HEVCDL_WEIGHT_FLOATS, wpath.c_str()); return 2; }
    fclose(fw); }

  // one block of frames per device entry
  struct PicOut { std::vector<uint8_t> bytes; size_t au_len = 0; char md5_text[128]; unsigned long long sse[3]; hevcdl_status st = HEVCDL_OK; };
  struct Shard {
    int dev = 0; long f_lo = 0, f_hi = 0; hevcdl_ctx *ctx = nullptr; int batch = 1; int rc = 0;
    double t_read = 0, t_dev = 0, t_host = 0, t_write = 0;
    std::vector<unsigned long long> rows;       // per coded picture: { poc, bits, sse Y, sse U, sse V, ctus, encode ns, device }  (what the devices gather)
    std::vector<std::vector<uint8_t>> aus;      // multi-device runs: the access units (+ hash SEI) of the block, written out in POC order at the end
    std::vector<std::string> md5;
  };
  enum { ROW = 8 };
  std::vector<Shard> shards(n_shards);
  for (int i = 0; i < n_shards; i++) { shards[i].dev = devices[i]; shards[i].f_lo = (long)i * n_frames / n_shards; shards[i].f_hi = (long)(i + 1) * n_frames / n_shards; }
  for (int i = 0; i < n_shards; i++) { // contexts: created one after the other (a refused allocation halves the batch of every shard)
    for (;;) {
      cfg.max_frames = batch; cfg.device = shards[i].dev;
      st = hevcdl_create(&cfg, weights.data(), weights.size(), &shards[i].ctx);
      // the decision kernel's workspace (up to 4.3 GB) is otherwise allocated by the first launch: reserved here, a lack of memory is met by the retry below
      bool ws_oom = false;
      if (st == HEVCDL_OK && (st = hevcdl_reserve_workspace(shards[i].ctx)) != HEVCDL_OK) { ws_oom = st == HEVCDL_ERR_OOM; hevcdl_destroy(shards[i].ctx); shards[i].ctx = nullptr; }
      if (st == HEVCDL_ERR_OOM && ws_oom) {
        // the workspace is sized by the device's CUs, not by the batch: a smaller batch does not shrink it.  What does: the independent launch form (a block per wave of
        // the context's own frames instead of every CU's workgroup), then the eight-wave build
        if (!(cfg.exec_flags & HEVCDL_EXEC_NO_UNIT_HANDOVER)) cfg.exec_flags |= HEVCDL_EXEC_NO_UNIT_HANDOVER;
        else if (!(cfg.exec_flags & HEVCDL_EXEC_RD_NARROW)) cfg.exec_flags = (cfg.exec_flags & ~HEVCDL_EXEC_RD_WIDE) | HEVCDL_EXEC_RD_NARROW;
        else { fprintf(stderr, "Error: device %d has no memory for the decision kernel's workspace (1.6 MB per wave of a launch) even in the independent launch form\n", shards[i].dev); return 3; }
        fprintf(stderr, "device %d: not enough memory for the decision kernel's workspace, retrying with exec_flags %d\n", shards[i].dev, (int)cfg.exec_flags);
        for (int j = 0; j < i; j++) { hevcdl_destroy(shards[j].ctx); shards[j].ctx = nullptr; }
        i = -1; break;
      }
      if (st != HEVCDL_ERR_OOM || batch == 1) break;
      batch = (batch + 1) / 2;
      fprintf(stderr, "device %d: not enough memory for the batch, retrying with %d pictures per call\n", shards[i].dev, batch);
      for (int j = 0; j < i; j++) { hevcdl_destroy(shards[j].ctx); shards[j].ctx = nullptr; }
      i = -1; break;
    }
    if (i < 0) continue;
    if (st == HEVCDL_ERR_INVALID_ARG) { fprintf(stderr, "Error: hevcdl_create rejected the configuration as invalid (e.g. tiles, which must be at least 4 CTUs wide and 1 CTU high: TComPicSym.cpp:380-392)\n"); return 2; }
    if (st != HEVCDL_OK) { fprintf(stderr, "Error: hevcdl_create failed on device %d with status %d (no GPU / unsupported configuration); there is no CPU path\n", shards[i].dev, (int)st); return 3; }
  }
  const double t_setup = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_main0).count();      // options, weights, contexts, the decision kernel's workspace
  const int chunk = (int)std::max<long>(1, std::min<long>(batch, opt.geti("ChunkFrames", 48)));

  printf("HEVC-DL MI355X path: %dx%d  QP %d  frames %ld (skip %ld)  batch %d  labels: %s\n", width, height, qp, n_frames, frame_skip, batch,
         label_dir.empty() ? (cnn_input == "luma" ? "on-device CNN (luma input)" : "on-device CNN (BT.601 RGB input)") : ("files under " + label_dir).c_str());
  if (multi) {
    printf("Devices:");
    for (int i = 0; i < n_shards; i++) printf(" %d (frames %ld..%ld)", shards[i].dev, shards[i].f_lo, shards[i].f_hi - 1);
    printf("\n");
  }
  if (!stage_keys.empty()) {
    printf("Accepted without effect on this path:");
    for (const auto &k : stage_keys) printf(" %s", k.c_str());
    printf("\n");
  }
  FILE *frec = recon_path.empty() ? nullptr : fopen(recon_path.c_str(), "wb");
  if (!recon_path.empty() && !frec) { fprintf(stderr, "Error: cannot open reconstruction file '%s'\n", recon_path.c_str()); return 2; }
  const std::string record_path = native_path(opt.get("RecordFile"));
  FILE *frecords = record_path.empty() ? nullptr : fopen(record_path.c_str(), "wb");
  FILE *fbits = bitstream_path.empty() ? nullptr : fopen(bitstream_path.c_str(), "wb");
  if (!bitstream_path.empty() && !fbits) { fprintf(stderr, "Error: cannot open bitstream file '%s'\n", bitstream_path.c_str()); return 2; }
  hevcdl_stream_config scfg; hevcdl_stream_config_default(&scfg, width, height, qp); scfg.level_idc = level_idc; scfg.sao_enabled = sao; scfg.tile_columns = tile_cols; scfg.tile_rows = tile_rows; scfg.bit_depth = bit_depth; scfg.wavefront = (int)wavefront;
  scfg.rewrite_param_sets = opt.geti("ReWriteParamSetsFlag", 1) != 0;
  scfg.tools = cfg.tools; scfg.lf_beta_offset_div2 = cfg.lf_beta_offset_div2; scfg.lf_tc_offset_div2 = cfg.lf_tc_offset_div2; scfg.loop_filter_disable = deblock ? 0 : 1;
  scfg.lf_across_tiles = cfg.lf_across_tiles; scfg.tile_uniform_spacing = cfg.tile_uniform_spacing; memcpy(scfg.tile_column_width, cfg.tile_column_width, sizeof scfg.tile_column_width); memcpy(scfg.tile_row_height, cfg.tile_row_height, sizeof scfg.tile_row_height);
  const double ny = (double)width * height, nc = ny / 4;
  double sum_bits = 0, sum_psnr[3] = { 0, 0, 0 }, sum_mse[3] = { 0, 0, 0 }; long done = 0;
  auto now = [] { return std::chrono::steady_clock::now(); };
  auto secs = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double>(b - a).count(); };
  // host threads for the per-picture work: the CPUs this process may really use (a container's CPU quota is often far below the node's thread count)
  unsigned cpus = std::max(1u, std::thread::hardware_concurrency());
  { FILE *fq = fopen("/sys/fs/cgroup/cpu.max", "r"); long long q = 0, per = 0; char qs[32];
    if (fq) { if (fscanf(fq, "%31s %lld", qs, &per) == 2 && strcmp(qs, "max") != 0 && per > 0) { q = atoll(qs); if (q > 0) cpus = (unsigned)std::max<long long>(1, std::min<long long>(cpus, q / per)); } fclose(fq); } }
  const int max_threads = (int)std::max(1u, std::min(cpus, 64u) / (unsigned)n_shards);
  const double maxval = (double)(255 << (bit_depth - 8));
  // one picture's log line and its share of the summary (TEncGOP.cpp:2500-2541; PSNR from the squared error as TEncGOP.cpp:2391-2394)
  auto log_picture = [&](long poc, unsigned long long bits, const unsigned long long *sse, double et, const char *md5_text) {
    const double p[3] = { psnr_of(sse[0], ny, maxval), psnr_of(sse[1], nc, maxval), psnr_of(sse[2], nc, maxval) };
    printf("POC %4ld TId: %1d ( %c-SLICE, QP %d ) %10llu bits [Y %6.4lf dB    U %6.4lf dB    V %6.4lf dB] [ET %5.0f ]%s\n", poc, 0, 'I', qp, bits, p[0], p[1], p[2], et, md5_text);
    sum_bits += (double)bits;
    for (int c = 0; c < 3; c++) { sum_psnr[c] += p[c]; sum_mse[c] += (double)sse[c] / (c ? nc : ny); }
    done++;
  };
  std::mutex file_mutex;                                  // reconstruction / record files: blocks are written at their own offsets
  // the app loop over one block of frames on one device: read -> device call -> per chunk of results: SSE of the output picture, the access unit (the arithmetic
  // coder), the picture hash on a pool of threads, then the output in POC order.  A single device writes as it goes; with several the access units and log rows
  // of a block are kept until every block is done.
  auto run_shard = [&](Shard &S) {
    FILE *fi = fopen(input.c_str(), "rb");
    if (!fi) { S.rc = 2; return; }
    const int sb = (int)std::min<long>(batch, S.f_hi - S.f_lo);
    // originals of one device call in page-locked memory (the upload then runs at the link's rate and does not go through a bounce buffer); label files beside them
    struct Pinned { uint8_t *p; explicit Pinned(size_t n) : p((uint8_t *)hevcdl_host_alloc(n)) {} ~Pinned() { hevcdl_host_free(p); } uint8_t *data() { return p; } } yuv_mem((size_t)frame_bytes * sb);
    if (!yuv_mem.data()) { fprintf(stderr, "Error: cannot allocate %zu bytes of page-locked memory for the originals\n", (size_t)frame_bytes * sb); S.rc = 3; fclose(fi); return; }
    std::vector<uint8_t> labels_mem(label_dir.empty() ? 0 : (size_t)ctus * 16 * sb);
    struct ChunkCtx { long f0; double et; const uint8_t *yuv; std::chrono::steady_clock::time_point t0; int nb; } cc = { 0, 0.0, nullptr, now(), 1 };
    auto on_chunk = [&](int first, int count, const hevcdl_ctu_record *recs, const void *pictures, const hevcdl_sao_blk *sao_params, const hevcdl_frame_stats *stats) -> int {
      const auto th0 = now();
      if (first == 0) cc.et = secs(cc.t0, th0) / cc.nb;                // device seconds per picture of this call (the log line's ET)
      const uint8_t *recon = (const uint8_t *)pictures;
      std::vector<PicOut> pics(count);
      {
        std::atomic<int> next(0);
        auto work = [&]() {
          std::vector<uint8_t> buf(hevcdl_access_unit_bound(width, height));
          for (int i = next++; i < count; i = next++) {
            PicOut &po = pics[i]; po.md5_text[0] = 0;
            for (int c = 0; c < 3; c++) po.sse[c] = stats[i].sse[c];
            if (deblock) { // the picture statistics follow the filtered picture: recomputed here
              const uint8_t *o = cc.yuv + frame_bytes * (size_t)(first + i), *r = recon + frame_bytes * (size_t)i;
              const size_t n[3] = { (size_t)width * height, (size_t)width * height / 4, (size_t)width * height / 4 };
              size_t off = 0;
              for (int c = 0; c < 3; c++) {
                unsigned long long sse = 0;
                if (bit_depth == 8) {
                  for (size_t k0 = 0; k0 < n[c]; k0 += 4096) { // 32-bit partial sums over short runs: the loop vectorises
                    unsigned part = 0; const size_t k1 = std::min(n[c], k0 + 4096);
                    for (size_t k = k0; k < k1; k++) { const int d = (int)o[off + k] - (int)r[off + k]; part += (unsigned)(d * d); }
                    sse += part;
                  }
                } else { const uint16_t *o16 = (const uint16_t *)o, *r16 = (const uint16_t *)r; for (size_t k = 0; k < n[c]; k++) { const int d = (int)o16[off + k] - (int)r16[off + k]; sse += (unsigned long long)(d * d); } }
                po.sse[c] = sse; off += n[c];
              }
            }
            // the access unit: VPS+SPS+PPS+slice, written to -b; its size is the picture's bit count (TEncGOP.cpp:2420-2447)
            po.st = hevcdl_write_access_unit(&scfg, (int)(cc.f0 + first + i), recs + (size_t)ctus * i, sao ? sao_params + (size_t)ctus * i : nullptr, buf.data(), buf.size(), &po.au_len);
            if (po.st != HEVCDL_OK) continue;
            po.bytes.assign(buf.begin(), buf.begin() + po.au_len);
            if (hash_sei) { // suffix SEI after the slice; not part of the picture's bit count (as in the reference)
              uint8_t sei[128], dg[48]; size_t sei_len = 0; int pb = 16;
              po.st = hevcdl_picture_hash(&scfg, recon + frame_bytes * (size_t)i, hash_sei, dg, &pb);
              if (po.st == HEVCDL_OK) po.st = hevcdl_write_hash_sei(hash_sei, dg, sei, sizeof sei, &sei_len);
              if (po.st != HEVCDL_OK) continue;
              po.bytes.insert(po.bytes.end(), sei, sei + sei_len);
              char *q = po.md5_text + sprintf(po.md5_text, " [%s:", hash_sei == 1 ? "MD5" : (hash_sei == 2 ? "CRC" : "Checksum"));          // TEncGOP.cpp:1160-1182
              for (int c = 0; c < 3; c++) { for (int k = 0; k < pb; k++) q += sprintf(q, "%02x", dg[pb * c + k]); *q++ = c < 2 ? ',' : ']'; }
              *q = 0;
            }
          }
        };
        const int nthreads = std::max(1, std::min(count, max_threads));
        std::vector<std::thread> pool;
        for (int t = 1; t < nthreads; t++) pool.emplace_back(work);
        work();
        for (auto &t : pool) t.join();
      }
      const auto tw0 = now();
      S.t_host += secs(th0, tw0);
      for (int i = 0; i < count; i++) {
        PicOut &po = pics[i];
        if (po.st != HEVCDL_OK) { fprintf(stderr, "Error: bitstream writer failed (status %d)\n", (int)po.st); S.rc = 3; return 1; }
        const long poc = cc.f0 + first + i;
        const unsigned long long row[ROW] = { (unsigned long long)poc, (unsigned long long)po.au_len * 8, po.sse[0], po.sse[1], po.sse[2], (unsigned long long)ctus, (unsigned long long)(cc.et * 1e9), (unsigned long long)S.dev };
        S.rows.insert(S.rows.end(), row, row + ROW);
        if (multi) { S.aus.push_back(std::move(po.bytes)); S.md5.push_back(po.md5_text); }
        else {
          if (fbits) fwrite(po.bytes.data(), 1, po.bytes.size(), fbits);
          log_picture(poc, (unsigned long long)po.au_len * 8, po.sse, cc.et, po.md5_text);
        }
      }
      if (frec || frecords) {
        std::lock_guard<std::mutex> lock(file_mutex);
        if (frec) { if (multi) fseeko(frec, (off_t)((cc.f0 + first) * (long long)frame_bytes), SEEK_SET); fwrite(recon, frame_bytes, count, frec); }
        if (frecords) { if (multi) fseeko(frecords, (off_t)((cc.f0 + first) * (long long)ctus * (long long)sizeof(hevcdl_ctu_record)), SEEK_SET); fwrite(recs, sizeof(hevcdl_ctu_record), (size_t)ctus * count, frecords); }
      }
      S.t_write += secs(tw0, now());
      return 0;
    };
    struct Tramp { static int call(void *u, int first, int count, const hevcdl_ctu_record *recs, const void *pics, const hevcdl_sao_blk *sp, const hevcdl_frame_stats *st)
                   { return (*(decltype(on_chunk) *)u)(first, count, recs, pics, sp, st); } };
    const long total = S.f_hi - S.f_lo, n_batches = (total + sb - 1) / sb;
    for (long bi = 0; bi < n_batches && S.rc == 0; bi++) {
      const long f0 = S.f_lo + bi * (long)sb; const int nb = (int)std::min<long>(sb, S.f_hi - f0);
      const auto tr0 = now();
      fseeko(fi, (off_t)((frame_skip + f0) * (long long)frame_bytes), SEEK_SET);
      if (fread(yuv_mem.data(), frame_bytes, nb, fi) != (size_t)nb) { fprintf(stderr, "Error: short read of '%s'\n", input.c_str()); S.rc = 2; break; }
      const uint8_t *lab = nullptr;
      if (!label_dir.empty()) {
        for (int i = 0; i < nb && S.rc == 0; i++) for (int a = 0; a < ctus; a++) {
          const std::string p = label_dir + "/" + std::to_string(f0 + i) + "/ctu" + std::to_string(a) + ".txt";
          std::ifstream lf(p); int v;
          for (int j = 0; j < 16; j++) { if (!(lf >> v) || v < 0 || v > 3) { fprintf(stderr, "Error: label file '%s' must hold 16 depths 0..3\n", p.c_str()); S.rc = 2; break; } labels_mem[((size_t)i * ctus + a) * 16 + j] = (uint8_t)v; }
          if (S.rc) break;
        }
        lab = labels_mem.data();
      }
      if (S.rc) break;
      const auto t0 = now();
      S.t_read += secs(tr0, t0);
      cc.f0 = f0; cc.yuv = yuv_mem.data(); cc.et = 0.0; cc.t0 = t0; cc.nb = nb;
      const double host_before = S.t_host + S.t_write;
      const hevcdl_status est = hevcdl_encode_pictures_chunked(S.ctx, yuv_mem.data(), nb, lab, deblock ? 1 : 0, sao ? 1 : 0, chunk, &Tramp::call, &on_chunk);
      S.t_dev += secs(t0, now()) - ((S.t_host + S.t_write) - host_before);
      if (est != HEVCDL_OK && S.rc == 0) { fprintf(stderr, "Error: %s (status %d)\n", hevcdl_last_error(S.ctx), (int)est); S.rc = 3; }
    }
    fclose(fi);
  };
  if (!multi) run_shard(shards[0]);
  else {
    std::vector<std::thread> th;
    for (int i = 1; i < n_shards; i++) th.emplace_back([&, i] { run_shard(shards[i]); });
    run_shard(shards[0]);
    for (auto &t : th) t.join();
  }
  int rc = 0;
  for (const Shard &S : shards) if (S.rc) rc = S.rc;
  if (multi && rc == 0) {
    // The per-picture rows of every device gathered with RCCL (north star: "RCCL over xGMI used only to gather per-frame rate / PSNR summaries"): one rank per
    // PHYSICAL device (ncclCommInitAll, a single process), every rank contributes the rows of its blocks padded to the largest contribution, one ncclAllGather; rank 0's
    // copy is the table the log and the summary are written from.
    std::vector<unsigned long long> table;
    std::string err, info;
    if (!gather_rows_rccl(shards.size(), [&](size_t i) { return shards[i].dev; }, [&](size_t i) -> const std::vector<unsigned long long> & { return shards[i].rows; }, ROW, table, err, &info)) {
      fprintf(stderr, "Error: gathering the per-picture rows with RCCL failed: %s\n", err.c_str()); rc = 3;
    } else {
      fprintf(stderr, "Picture rows gathered: %s\n", info.c_str());
      std::vector<const unsigned long long *> rows;
      for (size_t o = 0; o + ROW <= table.size(); o += ROW) if (table[o] != ~0ull) rows.push_back(&table[o]);
      std::sort(rows.begin(), rows.end(), [](const unsigned long long *a, const unsigned long long *b) { return a[0] < b[0]; });
      if ((long)rows.size() != n_frames) { fprintf(stderr, "Error: %zu rows gathered for %ld pictures\n", rows.size(), n_frames); rc = 3; }
      for (size_t r = 0; r < rows.size() && rc == 0; r++) { // POC order: blocks are contiguous, so this is the blocks one after the other
        const long poc = (long)rows[r][0];
        int si = 0; while (si + 1 < n_shards && poc >= shards[si].f_hi) si++;
        const Shard &S = shards[si];
        const size_t li = (size_t)(poc - S.f_lo);
        if (fbits) fwrite(S.aus[li].data(), 1, S.aus[li].size(), fbits);
        log_picture(poc, rows[r][1], rows[r] + 2, (double)rows[r][6] * 1e-9, S.md5[li].c_str());
      }
    }
  }
  { double t_read = 0, t_dev = 0, t_host = 0, t_write = 0;
    for (const Shard &S : shards) { t_read = std::max(t_read, S.t_read); t_dev = std::max(t_dev, S.t_dev); t_host = std::max(t_host, S.t_host); t_write = std::max(t_write, S.t_write); }
    fprintf(stderr, "stage seconds%s: set-up (contexts, workspace) %.2f  read %.2f  device (upload + CNN + decisions + filters; its chunk copies run behind the host work) %.2f  host (entropy coding, hashes, %d threads) %.2f  write %.2f  whole run %.2f\n",
            multi ? " (slowest device)" : "", t_setup, t_read, t_dev, max_threads, t_host, t_write, std::chrono::duration<double>(std::chrono::steady_clock::now() - t_main0).count()); }
  if (rc == 0 && done > 0) { // TEncAnalyze::printOut, 4:2:0 layout
    const double mse_yuv = (4 * sum_mse[0] + sum_mse[1] + sum_mse[2]) / done / 6.0;
    printf("\n\nSUMMARY --------------------------------------------------------\n");
    printf("\tTotal Frames |   Bitrate     Y-PSNR    U-PSNR    V-PSNR    YUV-PSNR  \n");
    printf("\t %8ld    %c %12.4lf  %8.4lf  %8.4lf  %8.4lf  %8.4lf  \n", done, 'a', sum_bits * (fps / 1000.0 / done), sum_psnr[0] / done, sum_psnr[1] / done,
           sum_psnr[2] / done, mse_yuv == 0 ? 999.99 : 10.0 * log10((double)(255 << (bit_depth - 8)) * (double)(255 << (bit_depth - 8)) / mse_yuv));
  }
  if (frec) fclose(frec);
  if (frecords) fclose(frecords);
  if (fbits) fclose(fbits);
  for (Shard &S : shards) hevcdl_destroy(S.ctx);
  return rc;
}
