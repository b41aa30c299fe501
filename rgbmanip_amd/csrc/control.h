// Argument records of the controller-step kernels (control.hip, synth_env.hip); mirrored field for field by
// rgbm_control_reward_args / rgbm_synth_scene in include/rgbm.h.
#pragma once
#include <hip/hip_runtime.h>

namespace rgbm {

struct ControlRewardArgs {
  const float* action;        // [N, lda] f32: xyz (3), dy, dz, unused, view weights (T)
  const double* cam_pose;     // [N,7]   env.camera_pose(robot_frame=True)
  const double* target;       // [N,7]   last_pose_target
  const float* move_success;  // [N]     cam_move_to()[0] as float32
  const double* bbox;         // [N,4]   bbox_queue[s % T]
  const double* avail;        // [N]     available[s % T]
  const double* gt_bbox;      // [N,8,3] gt_bbox[s]
  const double* pred_bbox;    // [N,8,3] pred_bbox[s]
  const double* pose_cur;     // [N,7]   pose_queue[s]
  const double* pose_prev;    // [N,7]   pose_queue[s-1]
  const double* robot_pose;   // [N,7]   env.robot_pose()
  const double* success;      // [N]
  double* reward;             // [N]
  double* terms;              // [17][N] or null
  double coef[14];            // cfg["reward"] in the order of the reward sum (rl_pose.py:319-334)
  double proper_pos[3];
  double precision2;          // precision ** 2 (0.01 for mugs, 0.04 otherwise)
  int N, T, lda, pots, first;
  int pad_;
};

struct SynthScene {
  const double* cam_pose;     // [N,7] camera pose in the robot frame (position, quaternion wxyz; x forward, y left, z up)
  const double* robot_pose;   // [N,7] robot root (position used; the robot frame is a pure translation of the world frame)
  const double* box;          // [N,15] handle box: centre (3), rows X, Y, Z of its axes in the world frame (9), half extents (3)
  double fx, fy, cx, cy;
  int N, H, W, env0;          // env0: global id of env 0 of this partition (background pattern)
};

int launch_control_reward(const ControlRewardArgs& a, hipStream_t s);
int launch_control_action(const float* action, int lda, const double* pose_mid, const double* pose_min, const double* pose_max,
                          double* pose, int N, hipStream_t s);
int launch_lookat_quat(const double* dir, double* quat, int N, int batch_zero, hipStream_t s);
int launch_control_grasp_frame(const double* est, double* center, double* direction, int N, hipStream_t s);
int launch_synth_camera(const SynthScene& sc, double* K, double* E, double* rays, hipStream_t s);
int launch_synth_render(const SynthScene& sc, const double* rays, float* color, unsigned char* mask, hipStream_t s);

}  // namespace rgbm
