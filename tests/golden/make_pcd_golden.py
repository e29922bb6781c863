"""Fixture for the PCD reader / writer: the header and the first and last points of one of the reference's own demo clouds
(/root/reference/demo/demo_data/pcd/000001.pcd, written by open3d with write_ascii=False), parsed here by hand so that the fixture does not
depend on this repo's reader.  Run in the build container (the reference tree is not on the GPU box)."""
import os

import numpy as np

SRC = "/root/reference/demo/demo_data/pcd/000001.pcd"
raw = open(SRC, "rb").read()
end = raw.index(b"DATA binary\n") + len(b"DATA binary\n")
header = raw[:end]
n = int([ln for ln in header.decode().splitlines() if ln.startswith("POINTS")][0].split()[1])
pts = np.frombuffer(raw, dtype="<f4", count=3 * n, offset=end).reshape(n, 3)
assert len(raw) == end + 12 * n
np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "pcd_sample.npz"), header=np.frombuffer(header, np.uint8), n_points=n,
                    head=pts[:256].copy(), tail=pts[-256:].copy(), checksum=np.float64(pts.astype(np.float64).sum(0)),
                    bytes_head=np.frombuffer(raw[end:end + 3072], np.uint8))
print("points", n)
