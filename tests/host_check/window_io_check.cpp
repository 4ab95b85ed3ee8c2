// Round trip of the window dump format through the C header: read <in>, write <out> (tests/test_window_io.py compares).
#include <cstdio>

#include "../../include/vilo_window_io.h"

int main(int argc, char **argv) {
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
