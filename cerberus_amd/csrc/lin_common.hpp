// Pieces of the linearisation pass that more than one translation unit needs: which buffer a pass writes, the per-lane view of a packed
// wave, and the second half of the frame-parallel visual form (k_visual_reduce's body: kernels_solve.hip launches it as a kernel of its
// own, kernels_asm_small.hip runs it in extra workgroups of k_assemble_s for small batches).
#pragma once
#include "solve_common.hpp"

#define LM_NTERM 21   // E, g, w_pose_s (6), w_ex0 (6), w_ex1 (6), w_td

// the LDS copy of the pose system the assembly kernels scatter into: packed lower triangle of the 80 x 80 system, row r at r (r + 1) / 2
#define CL_N 3240
__device__ __forceinline__ int cl_pos(int hi, int lo) { return ((hi * (hi + 1)) >> 1) + lo; }

// Linearisation modes. 0: at the current point (x, lambda), for windows that ask for it (need_lin) — the marginalisation's preMarginalize
// pass; the landmark gradients go to buffer 0. 1: at the candidate (xc, lambda_c) of every window still iterating — the solve loop; the
// landmark gradients go to the buffer the current linearisation does NOT use (k_accept flips st.cur when the candidate is accepted; the
// steps that follow a rejected candidate still need the current one's gradients). Everything else a linearisation writes is read only
// right after an accepted candidate and has one buffer.
__device__ __forceinline__ bool lin_skip(const SolverState &st, int mode) { return st.done || (mode == 0 && !st.need_lin); }
__device__ __forceinline__ double *lin_lm_g(BatchDev &b, const SolverState &st, int mode) { return mode ? b.lm_gbuf[1 - st.cur] : b.lm_gbuf[0]; }

// Per-lane view of a packed wave (WaveMeta): which chunk (start frame) a lane belongs to.
struct LaneSeg {
  int seg, s, gi, li;   // segment (-1: padding lane), start frame, global / window-local landmark index
  bool active;
};
__device__ __forceinline__ LaneSeg lane_segment(const WaveMeta &wv, const ChunkMeta *chunks, int lane, int cs[4], int cn[4], int ckm[4], int cgo[4]) {
  LaneSeg ls;
  ls.seg = -1; ls.s = 0; ls.gi = 0; ls.li = 0; ls.active = false;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    cs[g] = 0; cn[g] = 0; ckm[g] = 0; cgo[g] = 0;
    if (g < wv.nseg) {
      const ChunkMeta cm = chunks[wv.seg_chunk[g]];
      cs[g] = cm.s; cn[g] = cm.n; ckm[g] = cm.kmax; cgo[g] = cm.gram_off;
      const int i = lane - wv.seg_lane0[g];
      if (i >= 0 && i < cm.n) { ls.seg = g; ls.s = cm.s; ls.gi = cm.lm_off + i; ls.li = cm.lm_local + i; ls.active = true; }
    }
  }
  return ls;
}


// Second half of the TPAR form: per landmark, the (frame, camera) terms in the order the walking form adds them (frames ascending,
// left camera before right; an unobserved factor contributed +0.0).
// cur_of: the landmark-gradient buffer the current linearisation uses (SolverState::cur, or the snapshot of it the linearisation pass
// took — b.lin_cur — when this body runs beside the trust-region bookkeeping that flips it: small batches, k_assemble_s's extra workgroups)
__device__ __forceinline__ void visual_reduce_body(BatchDev &b, int wave_id, int mode, bool snapshot) {
  const WaveMeta wv = b.wave[wave_id];
  const SolverState &st = b.st[wv.win];
  if (lin_skip(st, mode)) return;
  const WinMeta wm = b.win[wv.win];
  const int lane = threadIdx.x & 63;
  int cs[4], cn[4], ckm[4], cgo[4];
  const LaneSeg ls = lane_segment(wv, b.chunk, lane, cs, cn, ckm, cgo);
  if (!ls.active) return;
  double acc[LM_NTERM];
#pragma unroll
  for (int v = 0; v < LM_NTERM; ++v) acc[v] = 0.0;
  for (int t = 0; t < wv.kmax; ++t)
    for (int cam = (t == 0 ? 1 : 0); cam < 2; ++cam) {
      const double *pt = b.lm_part + ((size_t)(t * 2 + cam) * LM_NTERM) * b.n_lm + ls.gi;
#pragma unroll
      for (int v = 0; v < LM_NTERM; ++v) acc[v] += pt[(size_t)v * b.n_lm];
    }
  double *wbase = b.lm_w + 80 * (size_t)wm.lm_off;
  const int L = wm.L, li = ls.li, s = ls.s;
  b.lm_E[ls.gi] = acc[0];
  (mode ? b.lm_gbuf[1 - (snapshot ? b.lin_cur[wv.win] : st.cur)] : b.lm_gbuf[0])[ls.gi] = acc[1];
  const bool ex_on = mode == 0 || !(wm.const_mask & CONST_EX), td_on = mode == 0 || !(wm.const_mask & CONST_TD);
  for (int c = 0; c < 6; ++c) {
    wbase[(size_t)(6 * s + c) * L + li] = acc[2 + c];
    wbase[(size_t)(CD_EX0 + c) * L + li] = ex_on ? acc[8 + c] : 0.0;
    wbase[(size_t)(CD_EX1 + c) * L + li] = ex_on ? acc[14 + c] : 0.0;
  }
  wbase[(size_t)CD_TD * L + li] = td_on ? acc[20] : 0.0;
  // the rows no frame's workgroup owns (like the walking form at its end; the rows of poses before the start frame are zero since
  // vilo_batch_create and nobody writes them)
  for (int f = s + wv.kmax; f < VILO_MAX_FRAMES; ++f)
    for (int c = 0; c < 6; ++c) wbase[(size_t)(6 * f + c) * L + li] = 0.0;
  wbase[(size_t)79 * L + li] = 0.0;
}
