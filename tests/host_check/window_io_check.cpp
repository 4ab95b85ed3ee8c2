// Round trip of the window dump format through the C header: read <in>, write <out> (tests/test_window_io.py compares).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/vilo_window_io.h"

// --fuzz <in> <scratch> <n>: n seeded mutations of <in> (bytes overwritten, runs zeroed / set to 0xff, truncations, a 32-bit field multiplied) are
// written to <scratch> and read back: every one must come back as a window or as an error code (the test builds this with the sanitizers).
static int fuzz(const char *in, const char *scratch, int n) {
  std::FILE *f = std::fopen(in, "rb");
  if (!f) return 2;
  std::vector<unsigned char> raw;
  unsigned char buf[4096];
  size_t k;
  while ((k = std::fread(buf, 1, sizeof buf, f)) > 0) raw.insert(raw.end(), buf, buf + k);
  std::fclose(f);
  unsigned long long s = 0x9e3779b97f4a7c15ull;
  auto rnd = [&](unsigned long long m) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return m ? s % m : 0; };
  int ok = 0, bad = 0;
  for (int it = 0; it < n; ++it) {
    std::vector<unsigned char> b = raw;
    const int edits = 1 + (int)rnd(3);
    for (int e = 0; e < edits; ++e) {
      size_t at = (size_t)rnd(rnd(2) ? (b.size() < 2048 ? b.size() : 2048) : b.size());   // (half of the edits in the headers)
      switch (rnd(5)) {
        case 0: b[at] = (unsigned char)rnd(256); break;
        case 1: for (size_t q = at; q < b.size() && q < at + 1 + rnd(64); ++q) b[q] = 0; break;
        case 2: for (size_t q = at; q < b.size() && q < at + 1 + rnd(16); ++q) b[q] = 0xff; break;
        case 3: b.resize(at < 8 ? 8 : at); break;
        default:
          if (at + 4 <= b.size()) {
            unsigned v; std::memcpy(&v, &b[at], 4);
            v *= (unsigned)(2 + rnd(65534)); std::memcpy(&b[at], &v, 4);
          }
      }
    }
    f = std::fopen(scratch, "wb");
    if (!f || std::fwrite(b.data(), 1, b.size(), f) != b.size()) return 2;
    std::fclose(f);
    vilo_window_file wf;
    if (vilo_window_read(scratch, &wf) == 0) { ++ok; vilo_window_free(&wf); } else ++bad;
  }
  std::printf("fuzz: %d read, %d refused\n", ok, bad);
  return bad > 0 ? 0 : 1;
}

int main(int argc, char **argv) {
  if (argc >= 5 && std::string(argv[1]) == "--fuzz") return fuzz(argv[2], argv[3], std::atoi(argv[4]));
  if (argc < 3) return 2;
  vilo_window_file wf;
  const int rc = vilo_window_read(argv[1], &wf);
  if (rc != 0) { std::printf("read failed %d\n", rc); return 1; }
  const int wrc = vilo_window_write(argv[2], &wf.cfg, &wf.desc, &wf.before, wf.has_after ? &wf.after : nullptr,
                                    wf.has_after ? wf.ref_summary : nullptr, wf.marginalization_flag);
  std::printf("frames %d landmarks %d obs %d prior %d after %d\n", wf.desc.n_frames, wf.desc.n_landmarks, wf.desc.n_obs,
              wf.desc.prior ? wf.desc.prior->n : 0, wf.has_after);
  vilo_window_free(&wf);
  return wrc == 0 ? 0 : 1;
}
