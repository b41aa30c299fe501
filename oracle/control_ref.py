"""CPU restatement (numpy) of the view queue of the reference's ControlInterface — TEST INFRASTRUCTURE ONLY.

Follows /root/reference/models/controller/rl_pose.py:
  reset_queue :89-101, add_view :118-149, add_bbox :151-155, get_state :157-171, get_observation :173-187,
  get_estimation :189-223 (view selection + the mugs corner permutation).
Pinned by tests/test_oracle_golden.py against tests/golden/control.npz, which tools/make_goldens.py produced by running
the reference class itself on the seeded view stream of rgbmanip_amd.synth.control_view.
The camera frame size is the reference's CAMERA_INTRINSIC[-1], [-2] = 480, 640.
"""
import numpy as np

H_IMG, W_IMG = 480, 640


class ControlQueueRef:
    def __init__(self, num_envs, max_steps, estimator=None):
        self.num_envs, self.max_steps, self.estimator = num_envs, max_steps, estimator
        self.reset_queue()

    def reset_queue(self):
        T, N = self.max_steps, self.num_envs
        self.image_queue = np.zeros((T, N, H_IMG, W_IMG, 3), dtype=np.float32)   # the reference keeps float64; values are copies
        self.mask_queue = np.zeros((T, N, H_IMG, W_IMG))
        self.bbox_queue = np.zeros((T, N, 4))
        self.pose_queue = np.zeros((T, N, 7))
        self.intrinsic_queue = np.zeros((T, N, 3, 3))
        self.extrinsic_queue = np.zeros((T, N, 4, 4))
        self.available = np.zeros((T, N))
        self.pred_bbox = np.zeros((T, N, 8, 3))
        self.gt_bbox = np.zeros((T, N, 8, 3))
        self.available_num = np.zeros((N,), dtype=np.int32)
        self.accumulate_steps = 0

    def add_view(self, image, cam_pose):
        k = self.accumulate_steps % self.max_steps
        cam = image["camera0"]
        self.image_queue[k] = cam["Color"]
        self.mask_queue[k] = cam["Mask"]
        self.pose_queue[k] = cam_pose
        self.intrinsic_queue[k] = cam["Intrinsic"]
        self.extrinsic_queue[k] = cam["Extrinsic"]
        p_env, p_x, p_y = np.nonzero(cam["Mask"])
        for i in range(self.num_envs):
            if p_env.shape[0]:                       # NB: "any env has a pixel", not "env i has a pixel" (rl_pose.py:132)
                x_min = np.min(np.where(p_env == i, p_x, H_IMG * 2))
                x_max = np.max(np.where(p_env == i, p_x, 0))
                y_min = np.min(np.where(p_env == i, p_y, W_IMG * 2))
                y_max = np.max(np.where(p_env == i, p_y, 0))
                self.available[k, i] = 1
                self.available_num[i] += 1
            else:
                x_min, x_max, y_min, y_max = H_IMG * 2, 0, W_IMG * 2, 0
                self.available[k, i] = 0
            self.bbox_queue[k, i] = [x_min / H_IMG, y_min / W_IMG, x_max / H_IMG, y_max / W_IMG]

    def add_bbox(self, pred_bbox, gt_bbox):
        k = self.accumulate_steps % self.max_steps
        self.pred_bbox[k] = pred_bbox
        self.gt_bbox[k] = gt_bbox

    def _time(self):
        t = np.zeros((self.num_envs, self.max_steps), dtype=np.float32)
        t[:, self.accumulate_steps - 1] = 1.0
        return t

    def get_state(self):
        centre = (self.gt_bbox[:, :, 0] + self.gt_bbox[:, :, 6]) / 2
        cur = np.concatenate((self.pose_queue, self.bbox_queue, centre), axis=-1).astype(np.float32)
        return np.concatenate((np.transpose(cur, (1, 0, 2)).reshape(self.num_envs, -1), self._time()), axis=-1)

    def get_observation(self):
        cur = np.concatenate((self.pose_queue, self.bbox_queue), axis=-1).astype(np.float32)
        return np.concatenate((np.transpose(cur, (1, 0, 2)).reshape(self.num_envs, -1), self._time()), axis=-1)

    def select_views(self):
        """The loop of get_estimation (:196-208): slot used%2 is overwritten by every available view in queue order."""
        N = self.num_envs
        K = np.zeros((2, N, 3, 3)); E = np.zeros((2, N, 4, 4))
        rgb = np.zeros((2, N, H_IMG, W_IMG, 3), dtype=np.float32); mask = np.zeros((2, N, H_IMG, W_IMG))
        used = np.zeros((N,), dtype=np.int32)
        for i in range(self.max_steps):
            for j in range(N):
                if self.available[i, j]:
                    s = used[j] % 2
                    K[s, j] = self.intrinsic_queue[i, j]; E[s, j] = self.extrinsic_queue[i, j]
                    rgb[s, j] = self.image_queue[i, j]; mask[s, j] = self.mask_queue[i, j]
                    used[j] += 1
        return K, rgb, mask, E

    def get_estimation(self):
        K, rgb, mask, E = self.select_views()
        bbox = self.estimator.estimate(K[0], rgb[0], mask[0], E[0], rgb[1], mask[1], E[1])
        if self.estimator.cfg["task_name"] == "mugs":
            bbox = bbox[:, [0, 2, 4, 6, 1, 3, 5, 7]]
        return bbox
