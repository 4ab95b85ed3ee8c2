"""ctypes mirrors of include/vilo_gpu.h / include/vilo_synth.h (plain C-ABI structs, no torch types)."""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_DIR = os.path.join(HERE, "lib")

F = 11  # VILO_MAX_FRAMES
MAX_PRIOR_BLOCKS = 40
MAX_PRIOR_DIM = 96

c_double_p = C.POINTER(C.c_double)
c_int32_p = C.POINTER(C.c_int32)
c_uint8_p = C.POINTER(C.c_uint8)


class Config(C.Structure):
    _fields_ = [
        ("acc_n", C.c_double), ("acc_n_z", C.c_double), ("acc_w", C.c_double), ("gyr_n", C.c_double), ("gyr_w", C.c_double),
        ("g_norm", C.c_double), ("phi_n", C.c_double), ("dphi_n", C.c_double), ("rho_c_n", C.c_double), ("rho_nc_n", C.c_double),
        ("v_n_min_xy", C.c_double), ("v_n_min_z", C.c_double), ("v_n_min", C.c_double), ("v_n_max", C.c_double),
        ("v_n_force_thres_ratio", C.c_double), ("v_n_term1_steep", C.c_double), ("v_n_term2_var_rescale", C.c_double),
        ("v_n_term3_distance_rescale", C.c_double), ("contact_sensor_type", C.c_int32), ("pad0", C.c_int32),
        ("rho_fix", C.c_double * 16), ("p_br", C.c_double * 3), ("R_br", C.c_double * 9),
        ("focal_length", C.c_double), ("huber_delta", C.c_double),
    ]


class Sample(C.Structure):
    _fields_ = [("dt", C.c_double), ("acc", C.c_double * 3), ("gyr", C.c_double * 3), ("phi", C.c_double * 12),
                ("dphi", C.c_double * 12), ("c", C.c_double * 4)]


SAMPLE_DOUBLES = 35
assert C.sizeof(Sample) == 8 * SAMPLE_DOUBLES


class Preint(C.Structure):
    _fields_ = [("sum_dt", C.c_double), ("delta_p", C.c_double * 3), ("delta_q", C.c_double * 4), ("delta_v", C.c_double * 3),
                ("delta_eps", C.c_double * 12), ("lin_ba", C.c_double * 3), ("lin_bg", C.c_double * 3), ("lin_rho", C.c_double * 4),
                ("jacobian", C.c_double * 961), ("covariance", C.c_double * 961)]


PREINT_DOUBLES = 33 + 2 * 961
assert C.sizeof(Preint) == 8 * PREINT_DOUBLES


class PreintImu(C.Structure):
    _fields_ = [("sum_dt", C.c_double), ("delta_p", C.c_double * 3), ("delta_q", C.c_double * 4), ("delta_v", C.c_double * 3),
                ("lin_ba", C.c_double * 3), ("lin_bg", C.c_double * 3), ("jacobian", C.c_double * 225), ("covariance", C.c_double * 225)]


PREINT_IMU_DOUBLES = 17 + 2 * 225
assert C.sizeof(PreintImu) == 8 * PREINT_IMU_DOUBLES


class Prior(C.Structure):
    _fields_ = [("n", C.c_int32), ("n_blocks", C.c_int32), ("block_id", C.c_int32 * MAX_PRIOR_BLOCKS),
                ("block_size", C.c_int32 * MAX_PRIOR_BLOCKS), ("block_idx", C.c_int32 * MAX_PRIOR_BLOCKS),
                ("x0", c_double_p), ("J0", c_double_p), ("r0", c_double_p), ("valid", C.c_int32), ("pad", C.c_int32)]


class WindowDesc(C.Structure):
    _fields_ = [("n_frames", C.c_int32), ("n_landmarks", C.c_int32), ("n_obs", C.c_int32), ("use_leg", C.c_int32),
                ("lm_start_frame", c_int32_p), ("lm_obs_offset", c_int32_p), ("obs", c_double_p), ("obs_is_stereo", c_uint8_p),
                ("preint", C.POINTER(Preint)), ("preint_imu", C.POINTER(PreintImu)), ("prior", C.POINTER(Prior)),
                ("leg_bias_const", C.c_int32), ("ex_const", C.c_int32), ("td_const", C.c_int32), ("pad", C.c_int32)]


class WindowState(C.Structure):
    _fields_ = [("pose", c_double_p), ("speed_bias", c_double_p), ("leg_bias", c_double_p), ("ex_pose", c_double_p),
                ("td", c_double_p), ("inv_depth", c_double_p)]


class SolveOpts(C.Structure):
    _fields_ = [("max_num_iterations", C.c_int32), ("fixed_iterations", C.c_int32),
                ("initial_trust_region_radius", C.c_double), ("max_trust_region_radius", C.c_double),
                ("min_trust_region_radius", C.c_double), ("min_relative_decrease", C.c_double),
                ("function_tolerance", C.c_double), ("gradient_tolerance", C.c_double), ("parameter_tolerance", C.c_double),
                ("min_lm_diagonal", C.c_double), ("max_lm_diagonal", C.c_double), ("jacobi_scaling", C.c_int32),
                ("max_solver_time_us", C.c_int32)]


class SolveSummary(C.Structure):
    _fields_ = [("iterations", C.c_int32), ("num_successful", C.c_int32), ("termination", C.c_int32), ("pad", C.c_int32),
                ("initial_cost", C.c_double), ("final_cost", C.c_double), ("cost_trace", C.c_double * 64),
                ("radius_trace", C.c_double * 64)]


class SynthParams(C.Structure):
    _fields_ = [("seed", C.c_uint64), ("n_landmarks", C.c_int32), ("n_start_frames", C.c_int32), ("imu_rate_hz", C.c_double),
                ("frame_rate_hz", C.c_double), ("pixel_noise", C.c_double), ("sig_p", C.c_double), ("sig_theta", C.c_double),
                ("sig_v", C.c_double), ("sig_ba", C.c_double), ("sig_bg", C.c_double), ("sig_rho", C.c_double),
                ("sig_lambda_rel", C.c_double), ("lin_offset_ba", C.c_double), ("lin_offset_bg", C.c_double),
                ("lin_offset_rho", C.c_double), ("with_prior", C.c_int32), ("pad", C.c_int32)]


class SynthOut(C.Structure):
    _fields_ = [("lm_start_frame", c_int32_p), ("lm_obs_offset", c_int32_p), ("obs", c_double_p), ("obs_is_stereo", c_uint8_p),
                ("samples", C.POINTER(Sample)), ("sample_offsets", c_int32_p), ("lin", c_double_p),
                ("pose", c_double_p), ("speed_bias", c_double_p), ("leg_bias", c_double_p), ("ex_pose", c_double_p),
                ("td", c_double_p), ("inv_depth", c_double_p),
                ("truth_pose", c_double_p), ("truth_speed_bias", c_double_p), ("truth_leg_bias", c_double_p),
                ("truth_inv_depth", c_double_p), ("prior", C.POINTER(Prior))]


def dptr(a):
    assert a.dtype == np.float64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(c_double_p)


def iptr(a):
    assert a.dtype == np.int32 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(c_int32_p)


def u8ptr(a):
    assert a.dtype == np.uint8 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(c_uint8_p)


def as_struct_ptr(a, ctype):
    """View a float64 numpy array as an array of `ctype` structs."""
    assert a.dtype == np.float64 and a.flags["C_CONTIGUOUS"]
    return C.cast(a.ctypes.data, C.POINTER(ctype))
