"""CPU tests of the host side: the C ABI library loads and exports every symbol include/fsg_hip.h
declares (no compute calls without a GPU), the module mirrors keep the reference's constructor / config /
state_dict contract, BASELINE config 1 (PointNet, the reference's CPU case) matches its golden vector, and
the HIP-only modules refuse CPU tensors instead of falling back."""
import os
import re

import numpy as np
import pytest
import torch

from golden_util import cloud, fill_state_dict, load

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from fissure_segmentation_amd import _lib
    header = open(os.path.join(ROOT, "include", "fsg_hip.h")).read()
    declared = set(re.findall(r"\b(fsg_[a-z0-9_]+)\s*\(", header))
    assert declared, "no declarations parsed"
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    for name in declared:
        assert hasattr(_lib.lib, name), name
    assert _lib.lib.fsg_version() >= 100
    assert isinstance(_lib.lib.fsg_last_error(), bytes)


def test_bad_arguments_are_reported_not_thrown():
    """argument validation happens on the host before any launch, so it is testable without a GPU"""
    from fissure_segmentation_amd import _lib
    with pytest.raises(RuntimeError, match="NULL pointer"):
        _lib.call("fsg_knn_dense_f32", None, 1, 16, 48, 16, 3, 4, 0, None, None, None, None)
    with pytest.raises(RuntimeError, match="k="):
        _lib.call("fsg_knn_dense_f32", 1, 1, 16, 48, 16, 3, 65, 0, 1, None, None, None)
    with pytest.raises(RuntimeError, match="NULL pointer"):
        _lib.call("fsg_knn_dense_ws_f32", None, 1, 2048, 3 * 2048, 2048, 3, 20, 0, None, None, None, 0, None)
    with pytest.raises(RuntimeError, match="workspace"):   # a workspace smaller than the (B,N) squared norms
        _lib.call("fsg_knn_dense_ws_f32", 1, 1, 2048, 3 * 2048, 2048, 3, 20, 0, 1, None, 1, 16, None)


def test_knn_workspace_query():
    """fsg_knn_dense_workspace_bytes: never less than the (B,N) fp32 squared norms the two-phase kernel needs; inside the
    coarse-sweep kernel's envelope also the point-major copy, the operand image and the (B, Np, Np / 32) survivor bitmaps that
    travel from the nominate launch to the refine launch (pure host arithmetic: no GPU needed)"""
    from fissure_segmentation_amd import _lib
    ws = _lib.lib.fsg_knn_dense_workspace_bytes
    assert ws(0, 2048, 3) == 0 and ws(8, 0, 3) == 0
    assert ws(8, 256, 3) == 8 * 256 * 4                     # below the envelope: the norms only
    assert ws(8, 2048, 200) == 8 * 2048 * 4                 # above 128 channels
    for B, N, C in ((8, 2048, 3), (8, 2048, 64), (4, 8192, 64), (8, 4096, 128), (2, 1500, 24)):
        Np = (N + 63) // 64 * 64
        cp = 4 if C <= 4 else 16 * (1 if C <= 16 else 2 if C <= 32 else 4 if C <= 64 else 8)
        bitmaps = B * Np * ((Np // 32 + 7) // 8 * 8) * 4            # round 4: survivor bitmaps between the two launches
        assert ws(B, N, C) >= B * Np * 4 * (1 + cp) + bitmaps, (B, N, C)   # norms + point-major fp32 copy + bitmaps at least
        assert ws(B, N, C) < 3 * B * Np * 4 * (2 + cp) + bitmaps + 8192, (B, N, C)


def test_no_cpu_fallback():
    import fissure_segmentation_amd as fsg
    from fissure_segmentation_amd.losses.chamfer_loss import ChamferLoss
    from fissure_segmentation_amd.models.dgcnn import DGCNNSeg
    from fissure_segmentation_amd.utils.general_utils import knn
    x = torch.randn(1, 3, 32)
    with pytest.raises(RuntimeError, match="GPU"):
        knn(x, 4)
    with pytest.raises(RuntimeError, match="GPU"):
        DGCNNSeg(k=4, in_features=3, num_classes=2)(x)
    with pytest.raises(RuntimeError, match="GPU"):
        ChamferLoss()(x, x)
    with pytest.raises(RuntimeError, match="GPU"):
        fsg.functional.edge_features(x, torch.zeros(1, 32, 4, dtype=torch.int32))


def test_module_contract_matches_reference_keys():
    from fissure_segmentation_amd.models.access_models import get_point_seg_model_class
    from fissure_segmentation_amd.models.folding_net import DGCNNFoldingNet
    for fixture, name, kw in [("dgcnnseg_dyn", "DGCNN", dict(in_features=3)),
                              ("dgcnnseg_stn", "DGCNN", dict(in_features=3, spatial_transformer=True)),
                              ("dgcnnseg_img", "DGCNN", dict(in_features=9, image_feat_module=True)),
                              ("pointnet_c1", "PointNet", dict(in_features=3))]:
        cls = get_point_seg_model_class(name)
        net = cls(num_classes=4, k=8, **kw)
        assert list(net.state_dict().keys()) == [str(s) for s in load(fixture)["keys"]], fixture
        clone = type(net)(**net.config)                 # train.py:505 of the reference
        assert clone.config == net.config
    for fixture, kw in [("ae_fold", {}), ("ae_deform_static", dict(deform=True, static=True))]:
        net = DGCNNFoldingNet(k=8, n_embedding=64, shape_type="plane", n_input_points=2048, decode_mesh=False, **kw)
        assert sorted(net.state_dict().keys()) == sorted(str(s) for s in load(fixture)["keys"])
    with pytest.raises(NotImplementedError):
        get_point_seg_model_class("nope")
    with pytest.raises(ValueError):
        get_point_seg_model_class("DGCNN")(k=4, in_features=3, num_classes=2, image_feat_module=True)


def test_save_load_roundtrip_cpu(tmp_path):
    from fissure_segmentation_amd.models.point_net import PointNetSeg
    net = PointNetSeg(5, 3).eval()
    path = str(tmp_path / "m.pth")
    net.save(path)
    assert set(torch.load(path).keys()) == {"config", "model_state"}
    back = PointNetSeg.load(path, "cpu").eval()
    x = torch.randn(2, 5, 40)
    assert torch.equal(net(x), back(x))


def test_pointnet_config1_cpu_vs_golden():
    """BASELINE configs[0]: PointNet seg, 8 synthetic 1024-point clouds, 4 classes, fp32 CPU."""
    from fissure_segmentation_amd.models.point_net import PointNetSeg
    g = load("pointnet_c1")
    net = fill_state_dict(PointNetSeg(3, 4), 501).train()
    x = torch.from_numpy(cloud(1501, 8, 3, 1024)).requires_grad_(True)
    y = net(x)
    gr = np.random.default_rng(2501).standard_normal(tuple(y.shape)).astype(np.float32)
    y.backward(torch.from_numpy(gr))
    np.testing.assert_allclose(y.detach().numpy(), g["logits"], rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(x.grad.numpy(), g["grad_x"], rtol=1e-3, atol=1e-4)
    for n, p in net.named_parameters():
        ref = float(g["gnorm_" + n])
        assert abs(float(p.grad.double().norm()) - ref) <= 1e-3 * ref + 1e-4, n


def test_plane_grids_match_reference_geometry():
    from fissure_segmentation_amd.shapes.shape_constructor import get_plane, get_plane_mesh
    from oracle import ref_cpu
    assert np.array_equal(get_plane(), ref_cpu.plane_grid_45())
    pts, faces = get_plane_mesh(n=16, xrange=(-0.3, 0.3), yrange=(-0.3, 0.3))
    assert pts.shape == (16, 2) and faces.shape == (18, 3)
    assert faces[:2].tolist() == [[0, 1, 4], [1, 4, 5]] and int(faces.max()) == 15


def test_flat_adam_equals_torch_adam():
    """optim.FlatAdam (one flat buffer, one fused kernel) == torch.optim.Adam over the separate tensors, step by step"""
    import copy
    from fissure_segmentation_amd.optim import FlatAdam
    torch.manual_seed(0)
    a = torch.nn.Sequential(torch.nn.Linear(5, 7), torch.nn.BatchNorm1d(7), torch.nn.ReLU(), torch.nn.Linear(7, 3))
    b = copy.deepcopy(a)
    oa, ob = torch.optim.Adam(a.parameters(), lr=1e-2), FlatAdam(b.parameters(), lr=1e-2)
    keys = list(b.state_dict().keys())
    for step in range(4):
        x = torch.randn(16, 5)
        for net, opt in ((a, oa), (b, ob)):
            opt.zero_grad(set_to_none=True)
            net(x).square().mean().backward()
            opt.step()
        for pa, pb in zip(a.parameters(), b.parameters()):
            torch.testing.assert_close(pa, pb, rtol=1e-6, atol=1e-7)
    assert list(b.state_dict().keys()) == keys


def test_install_reference_aliases_and_loss_registry(monkeypatch):
    """`install_reference_aliases()` puts this package's modules under the reference's import names, so the literal import
    lines of train.py / train_pc_ae.py / model_trainer.py (`from models.dgcnn import DGCNNSeg`, `from
    losses.access_losses import get_loss_fn`, ...) resolve to the HIP-backed classes; `get_loss_fn` follows
    losses/access_losses.py:43-93 (out-of-scope criteria raise NotImplementedError, unknown names ValueError)."""
    import sys
    import fissure_segmentation_amd as fsg
    saved = dict(sys.modules)
    try:
        fsg.install_reference_aliases()
        from models.dgcnn import DGCNNSeg, EdgeConv, SharedFullyConnected  # noqa: F401
        from models.point_net import PointNetSeg  # noqa: F401
        from models.pointtransformer.seg_model import PointTransformerCompatibility
        from models.folding_net import DGCNNFoldingNet  # noqa: F401
        from models.access_models import get_point_seg_model_class
        from models.modelio import LoadableModel
        from losses.access_losses import Losses, get_loss_fn
        from losses.chamfer_loss import ChamferLoss
        from losses.nnu_loss import NNULoss
        assert DGCNNSeg.__module__.startswith("fissure_segmentation_amd") or "fissure" in DGCNNSeg.__module__
        assert get_point_seg_model_class("DGCNN") is DGCNNSeg
        assert get_point_seg_model_class("PointTransformer") is PointTransformerCompatibility
        assert issubclass(DGCNNSeg, LoadableModel)
        crit = get_loss_fn("nnunet", torch.ones(4))
        assert isinstance(crit, NNULoss) and isinstance(get_loss_fn(Losses.NNUNET), NNULoss)
        assert isinstance(get_loss_fn("chamfer"), ChamferLoss)
        assert isinstance(get_loss_fn("ce", torch.ones(3)), torch.nn.CrossEntropyLoss)
        assert Losses.list() == ["nnunet", "ce", "recall", "ssm", "chamfer", "mesh", "dpsr"]
        for name in ("recall", "ssm", "dpsr"):
            with pytest.raises(NotImplementedError, match="outside the MI355X hot path"):
                get_loss_fn(name)
        # 'mesh' (losses/access_losses.py:67-77 of the reference): this package's RegularizedMeshLoss, Chamfer term on the HIP
        # kernel; default / explicit term weights as in the reference; the regularisers refuse to run without pytorch3d Meshes
        from losses.mesh_loss import RegularizedMeshLoss
        # (pytorch3d is absent here: a regulariser that cannot be served fails at CONSTRUCTION, where the reference's own import
        # would have failed -- not on the first forward after dataset and model are set up)
        with pytest.raises(NotImplementedError, match="need pytorch3d"):
            get_loss_fn("mesh")                 # the reference's default weights carry all three regularisers
        m = get_loss_fn("mesh", term_weights=[1., 0., 0., 0.])
        assert isinstance(m, RegularizedMeshLoss)
        assert (m.w_chamfer, m.w_edge_length, m.w_normal_consistency, m.w_laplacian, m.n_samples) == (1., 0., 0., 0., 2048)
        with pytest.raises(NotImplementedError, match="Laplacian"):
            get_loss_fn(Losses.MESH, term_weights=[2., 0., 0., 0.5])
        with pytest.raises(AssertionError):
            get_loss_fn("mesh", term_weights=[1., 1.])
        zero = RegularizedMeshLoss(0., 0., 0., 0.)
        assert zero(torch.zeros(1, 4, 3), torch.zeros(1, 4, 3)) == (0, {})
        with pytest.raises(NotImplementedError, match="edge length"):
            RegularizedMeshLoss(0., 1., 0., 0.)
        with pytest.raises(TypeError, match="cannot draw surface samples"):
            RegularizedMeshLoss(1., 0., 0., 0.)(object(), object())
        with pytest.raises(ValueError, match="No loss function named"):
            get_loss_fn("nope")
    finally:
        for k in list(sys.modules):
            if k not in saved:
                del sys.modules[k]
        sys.modules.update(saved)


def test_autograd_functions_leave_the_autocast_region():
    """every autograd.Function of the HIP path is wrapped by torch.amp.custom_fwd(cast_inputs=float32) / custom_bwd, so
    the reference's `with autocast(): model(x)` (model_trainer.py:157-161) cannot hand fp16 tensors to fp32 kernels"""
    import inspect
    from fissure_segmentation_amd import functional as F_hip
    n = 0
    for name, obj in vars(F_hip).items():
        if inspect.isclass(obj) and issubclass(obj, torch.autograd.Function) and obj is not torch.autograd.Function:
            n += 1
            assert hasattr(obj.forward, "__wrapped__") and hasattr(obj.backward, "__wrapped__"), name
    assert n >= 19


def test_zero_arena_and_deferral_rules_host_logic():
    """host-side bookkeeping of two round-4 mechanisms, no kernel involved: ZeroArena hands out disjoint, aligned, zeroed slices
    of ONE buffer (and fresh zeros for a second take / a late reservation); a weight gradient's split sum may be deferred only
    when it ends in leaf parameters with an empty `.grad`, outside create_graph, and the switch is on."""
    from fissure_segmentation_amd import functional as F_hip
    with F_hip.zero_arena() as ar:
        t1, t2 = F_hip._reserve_zeros(100), F_hip._reserve_zeros(7)
    assert t1[1] == 0 and t2[1] == 128 and ar.total == 192          # 64-element (256-byte) granules
    a = F_hip._take_zeros(t1, 100, "cpu")
    b = F_hip._take_zeros(t2, 7, "cpu")
    assert a.numel() == 100 and b.numel() == 7 and float(a.abs().sum() + b.abs().sum()) == 0.0
    assert a.data_ptr() + 4 * 128 == b.data_ptr()                    # slices of one buffer
    a.fill_(1.0)
    again = F_hip._take_zeros(t1, 100, "cpu")                        # second backward through the same graph: fresh zeros
    assert float(again.abs().sum()) == 0.0 and again.data_ptr() != a.data_ptr()
    assert ar.reserve(5) is None                                     # the buffer exists: a late forward gets its own zeros
    assert F_hip._reserve_zeros(10) is None and F_hip._take_zeros(None, 10, "cpu").numel() == 10      # outside the context
    w, b_ = torch.nn.Parameter(torch.zeros(3, 3)), torch.nn.Parameter(torch.zeros(3))
    derived = w * 1.0                                                # not a leaf: somebody's backward reads its gradient
    with torch.no_grad():
        assert F_hip._may_defer(F_hip._grad_targets(w) + F_hip._grad_targets(b_))
        assert F_hip._grad_targets(derived) is None and not F_hip._may_defer(F_hip._grad_targets(derived))
        w.grad = torch.zeros(3, 3)
        assert not F_hip._may_defer((w,))                            # AccumulateGrad would ADD to it during the pass
        w.grad = None
        F_hip.set_deferred_weight_grads(False)
        try:
            assert not F_hip._may_defer((w,))
        finally:
            F_hip.set_deferred_weight_grads(True)
    assert not F_hip._may_defer((w,))                                # grad mode on (create_graph): the engine clones
