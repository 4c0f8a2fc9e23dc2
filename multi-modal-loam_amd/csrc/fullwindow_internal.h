// fullwindow_internal.h -- the full-window solver handle shared by window_imu.hip (host trust-region loop, IMU factor,
// marginalization) and fullwindow_dev.hip (the same loop resident on the device).
#pragma once
#include <vector>

#include "../../include/mmloam_hip.h"

struct MmlFwPrior {       // MarginalizationFactor on (para_PR[0], para_VBias[0]) after the address shift (:1552-1562)
    bool valid = false;
    int nres = 15;
    double J[15 * 15];    // linearized_jacobians, columns [PR 6 | VBias 9]
    double r0[15];        // linearized_residuals
    double x0[15];        // keep_block_data
};

struct MmlFwEval {  // dense normal equations of the whole window at one x
    std::vector<double> H, g;
    double cost = 0;
};

struct mml_fullwindow {
    int W = 0, n = 0;
    mml_solve_opts opts;
    std::vector<mml_imu_preint> imu;   // imu[f]: between frame f-1 and f (f >= 1)
    std::vector<char> have_imu;
    std::vector<double> U;             // sqrt information of imu[f] (225 each), factored once in mml_fullwindow_set_imu
    std::vector<char> U_ok;
    double gravity[3] = {0, 0, 0};
    MmlFwPrior prior;
    // trust-region state (Ceres 2.1 TRADITIONAL_DOGLEG, same constants as mml_solve / tr_propose / tr_decide)
    std::vector<double> x, xc, x_init, scale, diag, grad, gn, step;
    MmlFwEval cur, cand;
    double radius = 1e4, mu = 1e-8, alpha = 0, dogleg_norm = 0, x_norm = 0, model_change = 0, step_norm = 0;
    int reuse = 0, num_invalid = 0, iter = 0, successful = 0, termination = 0, started = 0, done = 0;
    double initial_cost = 0;
};

// sqrt information of one pre-integration (15 x 15 upper, row-major): LLT(covariance^-1).matrixL().transpose()
// (Estimator.cpp:1240-1242); false when the covariance is not positive definite.  window_imu.hip.
bool mml_imu_sqrt_info(const mml_imu_preint* pre, double* U);
