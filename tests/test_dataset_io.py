"""Offline dataset format (datasetGenerator.py:76-124 writer, dataset.py:12-80 reader): round trip on the CPU."""
import pickle

import numpy as np
from PIL import Image

from occlusionenv_amd import dataset_io


def test_write_then_read_round_trip(tmp_path):
    rng = np.random.default_rng(0)
    S = 32
    for run in range(2):
        w = dataset_io.RunWriter(str(tmp_path), run)
        for j in range(3):
            obs = rng.random((4, S, S)).astype(np.float32)
            obs[3] = np.where(rng.random((S, S)) < 0.5, -1.0, 3.0 + obs[3])  # depth, -1 = background
            occl = rng.random((S, S)).astype(np.float32)
            w.write_frame(j, obs, occl, 0.1 * j, -0.2 * j, (1.0 + j, -2.0 - j))
        w.close()
    arr = pickle.load(open(tmp_path / "run_1" / "params.pickle", "rb"))
    assert arr.shape == (15,) and np.allclose(arr.reshape(-1, 5)[2], [2, 0.2, -0.4, 3.0, -4.0])
    d = np.asarray(Image.open(tmp_path / "run_0" / "Depth" / "0.png"))
    assert d.dtype == np.uint8 and d.min() == 0 and 153 <= d[d > 0].min() and d.max() <= 204  # depth in [3,4) * 51
    ds = dataset_io.OcclusionDataset(str(tmp_path), size=(16, 16))
    assert len(ds) == 6
    img, label, pos, grad = ds[4]
    assert img.shape == (4, 16, 16) and label.shape == (1, 16, 16) and float(img[:3].min()) >= 0 and float(img[:3].max()) <= 1
    assert np.allclose(pos.numpy(), [0.1, -0.2]) and np.allclose(grad.numpy(), [2.0, -3.0])
