"""CPU oracle for the SEE-VCN hot path — TEST INFRASTRUCTURE ONLY.

A CPU restatement (numpy / torch-CPU / plain C) of the reference's algorithms, each function citing the
reference file:line it follows.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
import or link anything under oracle/; the product (see-vcn_amd/) never does and fails loudly when the HIP
library is missing.

Pinning status:
  voxelize.py   pinned: tests/golden/dyn_voxel_*.npz, mean_vfe.npz come from the reference's own DynamicMeanVFE/MeanVFE
  vcn.py        pinned: tests/golden/vcn_vc.npz, vcn_cn.npz come from the reference's own VCN_VC/VCN_CN
  sparse conv   PARITY UNPINNED: spconv is an un-vendored, unpinned third-party dependency;
                the oracle restates its published semantics with a canonical output order.
"""
