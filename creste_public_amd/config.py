"""Config objects for the CREStE hot path.

The reference passes OmegaConf `DictConfig`s into every model constructor
(attribute access + `.get` + `[]`; /root/reference/creste/models/terrainnet.py:28-60).
OmegaConf/Hydra are not part of this image, so `Cfg` provides the same access
protocol over plain dicts; an OmegaConf DictConfig (if the caller has one) can be
passed straight in as well because only that protocol is used.

`terrainnet_cfg()` / `maxent_irl_cfg()` restate the hyper-parameters of the
reference's shipped model configs
(configs/model/ssc_sam/terrainnet_supcon_sam2dynelev_jointdinopretrain.yaml,
configs/model/traversability/terrainnet_maxentirlcf_msfcn_sam2dynsemelev.yaml).
"""
from __future__ import annotations

import copy
from typing import Any, Mapping


class Cfg(dict):
    """dict with attribute access, recursive wrapping and OmegaConf-like `.get`."""

    def __init__(self, data: Mapping | None = None, **kw):
        super().__init__()
        src = dict(data or {})
        src.update(kw)
        for k, v in src.items():
            self[k] = v

    @staticmethod
    def wrap(v: Any) -> Any:
        if isinstance(v, Cfg):
            return v
        if isinstance(v, Mapping):
            return Cfg(v)
        if isinstance(v, (list, tuple)):
            return [Cfg.wrap(x) for x in v]
        return v

    def __setitem__(self, k, v):
        super().__setitem__(k, Cfg.wrap(v))

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v

    def __deepcopy__(self, memo):
        return Cfg({k: copy.deepcopy(v, memo) for k, v in self.items()})

    def to_dict(self) -> dict:
        def un(v):
            if isinstance(v, Cfg):
                return {k: un(x) for k, x in v.items()}
            if isinstance(v, list):
                return [un(x) for x in v]
            return v
        return un(self)


def as_cfg(obj: Any) -> Cfg:
    """Accept a dict / Cfg / OmegaConf DictConfig and return a Cfg."""
    if isinstance(obj, Cfg):
        return obj
    if isinstance(obj, Mapping):
        return Cfg(obj)
    # OmegaConf DictConfig (duck-typed; not importable here)
    try:
        from omegaconf import OmegaConf  # type: ignore
        return Cfg(OmegaConf.to_container(obj, resolve=True))
    except Exception as e:  # pragma: no cover
        raise TypeError(f"cannot interpret {type(obj)} as a model config") from e


def terrainnet_cfg(image_size=(512, 612), views: int = 1) -> Cfg:
    """TerrainNet (RGB-D encoder + splat + BEV heads) hyper-parameters."""
    depth_embed_dim, fdn_embed_dim, num_bins = 256, 128, 128
    discretize = dict(mode="UD", num_bins=num_bins, depth_min=300, depth_max=25600)
    return Cfg(
        project_name="TerrainNetSAM",
        load_setting="strict",
        use_temporal=False,
        use_movability=False,
        multiview_distillation=False,
        depth_embed_dim=depth_embed_dim,
        fdn_embed_dim=fdn_embed_dim,
        num_depth_bins=num_bins,
        weights_path="",
        views=views,
        discretize=discretize,
        vision_backbone=dict(
            class_name="DistillationBackbone",
            name="efficientnet-b0",
            input_type="rgbd",
            weights_path="",
            return_feats=True,
            effnet_cfgs=dict(in_channels=4, out_channels=depth_embed_dim,
                             downsample=4, image_size=list(image_size)),
        ),
        camera_projector=dict(
            name="Cam2MapMulti",
            voxel_size=[0.1, 0.1, 3],
            point_cloud_range=[-12.8, -12.8, -2, 12.8, 12.8, 1],
            embed_z=True, z_embed_dim=32, z_embed_mode="mlp", num_cams=1,
            splat_key="depth_preds_feats",
            vision_fusion=dict(name="ConvEncoder", dims=[288, 96], kernels=[1],
                               paddings=[0], norm_type="batch_norm"),
        ),
        depth_head=dict(name="depthconv-head", dims=[depth_embed_dim, num_bins],
                        kernels=[3], paddings=[1], norm_type="batch_norm"),
        distillation_head=dict(
            name="distillation-head",
            feature_head=dict(name="MultiLayerConv", kernels=[1, 1, 1],
                              paddings=[0, 0, 0],
                              dims=[depth_embed_dim, 128, 128, fdn_embed_dim],
                              norm_type="batch_norm"),
        ),
        bev_classifier=dict(
            name="InpaintingResNet18MultiHead",
            net_kwargs=dict(input_key="bev_features", num_input_features=96,
                            num_classes=[32, 6, 2],
                            output_prefix=["inpainting_sam", "inpainting_sam_dynamic",
                                           "elevation"]),
        ),
    )


def maxent_irl_cfg(image_size=(512, 612), solve_mdp: bool = True,
                   map_size=(64, 128), map_ds: int = 2, point_cloud_range=None, voxel_size=None) -> Cfg:
    """MaxEntIRL (frozen TerrainNet + reward FCN + VI/SVF) hyper-parameters.  The MDP grid `map_size` must equal
    ((rows / map_ds) // 2, cols / map_ds) of the BEV grid (max-pool by map_ds, front-half crop: vin.py:104-109);
    the shipped config is a 256x256 BEV grid (0.1 m voxels over +-12.8 m), map_ds 2 -> 64x128."""
    backbone = terrainnet_cfg(image_size)
    if point_cloud_range is not None:
        backbone["camera_projector"]["point_cloud_range"] = list(point_cloud_range)
    if voxel_size is not None:
        backbone["camera_projector"]["voxel_size"] = list(voxel_size)
    backbone["load_setting"] = "strict_freeze"
    feats_dim = 40
    return Cfg(
        project_name="TraversabilityLearning",
        ckpt_path="", weights_path="", load_strict=True, freeze_weights=True,
        map_ds=map_ds, views=1, action_horizon=50, zero_terminal_state=False,
        policy_method="pp",
        policy_kwargs=dict(method="sharpen", temperature=0.005),
        solve_mdp=solve_mdp,
        map_size=list(map_size),
        vision_backbone=backbone,
        traversability_head=dict(
            name="MaxEntIRL", value_iterator="VIN", feats_dim=feats_dim, map_size=128,
            policy_method="pp",
            net_kwargs=dict(
                reward_cfg=dict(
                    name="MultiScaleFCN", ds=map_ds,
                    input_keys=["inpainting_sam_preds", "inpainting_sam_dynamic_preds",
                                "elevation_preds"],
                    output_prefix=["traversability_preds"],
                    net_kwargs=dict(
                        prepool=dict(dims=[feats_dim, 64, 32], kernels=[5, 3],
                                     stride=[1, 1], norm_type="batch_norm"),
                        skip=dict(dims=[32, 32, 16], kernels=[3, 1], stride=[1, 1],
                                  norm_type="batch_norm"),
                        trunk=dict(dims=[32, 32, 32], kernels=[3, 1], stride=[1, 1],
                                   norm_type="batch_norm"),
                        postpool=dict(dims=[48, 1], kernels=[1], stride=[1],
                                      norm_type="batch_norm"),
                    ),
                ),
                qvalue_cfg=dict(dims=[1, 8], kernels=[3], stride=[1], padding=[1],
                                input_keys=["traversability"], norm_type="batch_norm",
                                discount=0.99),
            ),
        ),
        batch_size=10,
        optimizer=dict(name="Adam", beta1=0.9, beta2=0.999, lr=0.0005),
        lr_scheduler=dict(name="ExponentialLR", gamma=0.96),
        loss=[dict(name="MaxEntIRLLoss", weight=1.0, map_ds=map_ds,
                   map_sz=list(map_size), maxent_weight=1.0, reward_weight=0.01,
                   alpha=0.5, use_fov_mask=True, pred_key="outputs/exp_svf",
                   fov_key="inputs/fov_mask", lab_key="inputs/traversability_label",
                   cf_key="inputs/counterfactuals_label")],
    )
