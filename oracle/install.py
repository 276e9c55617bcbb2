"""Context manager that points the host-side mirror at the CPU oracle (test infrastructure).

Used by tests, ``__graft_entry__.smoke()`` (as the checker) and ``bench.py``'s ``cpu_baseline``
leg only.  The product package never imports this."""
import contextlib


@contextlib.contextmanager
def oracle_backend():
    from . import pn2_ext_oracle, region_oracle
    import regnet_for_3d_grasping_amd.get_regiondataset as grd
    import regnet_for_3d_grasping_amd.gripper_region_network as grn
    import regnet_for_3d_grasping_amd.pn2_utils.function as fn
    import regnet_for_3d_grasping_amd.pn2_utils.functions.gather_knn as gk
    saved = (fn.pn2_ext, gk.dgcnn_ext, grd.region_ops, grn.region_ops)
    fn.pn2_ext, gk.dgcnn_ext, grd.region_ops, grn.region_ops = (pn2_ext_oracle, pn2_ext_oracle, region_oracle,
                                                                 region_oracle)
    try:
        yield pn2_ext_oracle
    finally:
        fn.pn2_ext, gk.dgcnn_ext, grd.region_ops, grn.region_ops = saved
