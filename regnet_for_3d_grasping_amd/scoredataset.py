"""Scene-record dataset: the reference's ``dataset_utils/scoredataset.py``.

A record is a pickled dict ``*.p`` (scoredataset.py:62-66; labels get_regiondataset.py:66-86):
  view_cloud (M,3), view_cloud_color (M,3), view_cloud_score (M,), view_cloud_label (M,)   -- the scene
  frame (G,4,4) + antipodal_score (G,)            or the ``select_*`` set                   -- the grasps
``ScoreDataset`` splits the record files 80/20 with numpy's legacy global RNG seeded by ``data_seed``
(:17-50), resamples every scene to exactly ``all_points_num`` points (without replacement when the record
has enough, :68-72), scales the colours of table and object points by random factors (:52-58) and squashes
the per-point score with tanh (:80).  The numpy RNG is consumed in exactly the reference's order, so a run
seeded like the reference's draws the same points (tests/test_scoredataset_cpu.py against fixtures produced
by the reference's class).
"""
import os
import pickle

import numpy as np
import torch.utils.data


def _complement(count, chosen):
    """The held-out files, in the iteration order of the reference's ``set(ori) - set(index)`` (:27-28,
    :47-48) -- CPython set order, which is NOT ascending once ``count`` exceeds the result's hash table."""
    return np.array(list(set(np.arange(count).tolist()) - set(np.asarray(chosen).tolist())))


class ScoreDataset(torch.utils.data.Dataset):
    """``ScoreDataset(all_points_num, path, tag, data_seed, data_width)`` -> items
    ``(view (N,6) float32 [xyz | rgb], tanh(score) (N,), label (N,), record path, width float32 array)``."""

    def __init__(self, all_points_num, path, tag, data_seed, data_width):
        self.all_points_num = all_points_num
        self.tag = tag
        self.width = np.array(data_width, dtype=np.float32)
        np.random.seed(data_seed)
        evaluation_set = "eval_data" in path
        if evaluation_set:
            self.base_path = path
        else:
            self.base_path = os.path.join(path, "training_data_test" if tag == "test" else "training_data")
        names = np.array(sorted(os.listdir(self.base_path)))
        if not evaluation_set and tag == "test":
            self.data_name = names                      # the test directory is used whole, no draw
            return
        index = np.random.choice(len(names), int(len(names) * 0.8), replace=False)
        held_out = (tag != "train") if evaluation_set else (tag == "validate")
        if held_out:
            index = _complement(len(names), index)
        self.data_name = names[index]

    def _noise_color(self, color, label):
        table_gain = np.random.rand(3)
        object_gain = 1 - np.random.rand(3) / 5
        table, objects = label == 0, label != 0
        for channel in range(3):
            color[table, channel] *= table_gain[channel]
            color[objects, channel] *= object_gain[channel]
        return color

    def __getitem__(self, index):
        data_path = os.path.join(self.base_path, self.data_name[index])
        data = np.load(data_path, allow_pickle=True)
        cloud = data["view_cloud"].astype(np.float32)
        color = data["view_cloud_color"].astype(np.float32)
        score = data["view_cloud_score"].astype(np.float32)
        label = data["view_cloud_label"].astype(np.float32)
        pick = np.random.choice(len(cloud), self.all_points_num, replace=len(cloud) < self.all_points_num)
        cloud, color, label, score = cloud[pick], color[pick], label[pick], score[pick]
        color = self._noise_color(color, label)
        return np.c_[cloud, color], np.tanh(score), label, data_path, self.width

    def gpu_item(self, index, device):
        """``__getitem__`` with everything after the file read on the GPU: the record's arrays go up once, the resampling
        positions and the six colour gains are drawn from numpy's stream BY THE DEVICE (np_random.choice_rows_device /
        rand_device: same values and same stream consumption as the host's np.random.choice + 2 x rand(3)), and one kernel
        gathers, jitters and squashes (csrc/dataset.hip).  Returns GPU tensors (view (N,6), tanh(score) (N,), label (N,))
        + the record path and width; call ``np_random.flush()`` (or leave a ``np_random.deferred()`` block) before using
        ``np.random`` on the host again."""
        import torch
        from . import _lib, np_random, region_ops
        data_path = os.path.join(self.base_path, self.data_name[index])
        data = np.load(data_path, allow_pickle=True)
        dev = torch.device(device)
        up = lambda key: torch.from_numpy(np.ascontiguousarray(data[key], dtype=np.float32)).to(dev, non_blocking=True)
        cloud, color, score, label = up("view_cloud"), up("view_cloud_color"), up("view_cloud_score"), up("view_cloud_label")
        M, N = cloud.shape[0], self.all_points_num
        counts = torch.full((1,), M, dtype=torch.int32, device=dev)
        # mode 0 = "without replacement when the list has at least `size` entries, else with" = scoredataset.py:68-72
        pick = np_random.choice_rows_device(counts, N, 0, M)[0].view(N)
        rand6 = np_random.rand_device(6, dev)
        with torch.cuda.device(dev):
            pc = torch.empty((N, 6), dtype=torch.float32, device=dev)
            score_out = torch.empty((N,), dtype=torch.float32, device=dev)
            label_out = torch.empty((N,), dtype=torch.float32, device=dev)
            _lib.check(_lib.lib.regnet_dataset_resample_f32(
                cloud.data_ptr(), color.data_ptr(), score.data_ptr(), label.data_ptr(), M, pick.data_ptr(), N,
                rand6.data_ptr(), pc.data_ptr(), score_out.data_ptr(), label_out.data_ptr(),
                region_ops._range_flag(dev).data_ptr(), torch.cuda.current_stream(dev).cuda_stream), "dataset_resample")
        return pc, score_out, label_out, data_path, self.width

    def gpu_batch(self, indices, device):
        """A batch of ``gpu_item``s stacked on the device: (pc (B,N,6), score (B,N), label (B,N), paths, widths) -- the
        collated batch of the reference's 8-worker DataLoader (utils.py:31-57) without the host-side resampling."""
        import torch
        from . import np_random
        with np_random.deferred():
            items = [self.gpu_item(i, device) for i in indices]
        return (torch.stack([it[0] for it in items]), torch.stack([it[1] for it in items]),
                torch.stack([it[2] for it in items]), [it[3] for it in items], np.stack([it[4] for it in items]))

    def __len__(self):
        return len(self.data_name)


def write_record(path, scene, score, label, grasps):
    """Write one record in the dataset's layout (a pickled dict, protocol 2 like the reference's files)."""
    scene = np.asarray(scene)
    record = {"view_cloud": scene[:, :3].astype(np.float32), "view_cloud_color": scene[:, 3:6].astype(np.float32),
              "view_cloud_score": np.asarray(score, np.float32), "view_cloud_label": np.asarray(label, np.float32)}
    record.update(grasps)
    with open(path, "wb") as f:
        pickle.dump(record, f, protocol=2)
