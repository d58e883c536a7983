"""MultiInputRCNN meta-architecture behind the reference's interface
(uwsod/detectron2/modeling/meta_arch/rcnn_multi.py:23-291; build_model meta_arch/build.py:15-23).

`model(batched_inputs) -> dict of 9 losses` in training, `list[{"instances": Instances}]` in eval; same
input dict keys (image{1,2}{,_flip} u8 CHW BGR, proposals*, instances*), 1 image per GPU (:148).
MI355X plan: the u8 -> normalised NHWC conversion is one fused kernel per view (sw_preprocess) instead of 4 H2D +
8 elementwise kernels, the two backbone calls keep the reference's batch-of-2 shape (view + flipped view)."""
import torch
import torch.nn as nn

from . import ops
from . import backbone_vgg, box_head, roi_heads_oicrplus  # noqa: F401  (registers the plugin entries)
from .registry import BACKBONE_REGISTRY, META_ARCH_REGISTRY, ROI_HEADS_REGISTRY
from .structures import ImageList, Instances


def build_backbone(cfg, input_shape=None):
    return BACKBONE_REGISTRY.get(cfg.MODEL.BACKBONE.NAME)(cfg, input_shape)


def build_roi_heads(cfg, input_shape):
    cls = ROI_HEADS_REGISTRY.get(cfg.MODEL.ROI_HEADS.NAME)
    return cls(**cls.from_config(cfg, input_shape))


def build_model(cfg):
    """meta_arch/build.py:15-23"""
    model = META_ARCH_REGISTRY.get(cfg.MODEL.META_ARCHITECTURE)(cfg)
    model.to(torch.device(cfg.MODEL.DEVICE))
    return model


@META_ARCH_REGISTRY.register()
class MultiInputRCNN(nn.Module):
    def __init__(self, cfg=None, *, backbone=None, proposal_generator=None, roi_heads=None, pixel_mean=None,
                 pixel_std=None, input_format="BGR", vis_period=0):
        super().__init__()
        if cfg is not None:
            backbone = build_backbone(cfg)
            roi_heads = build_roi_heads(cfg, backbone.output_shape())
            pixel_mean, pixel_std = cfg.MODEL.PIXEL_MEAN, cfg.MODEL.PIXEL_STD
            input_format, vis_period = cfg.INPUT.FORMAT, cfg.VIS_PERIOD
            assert cfg.MODEL.PROPOSAL_GENERATOR.NAME == "PrecomputedProposals"   # -> None (proposal_generator/build.py:20-22)
        self.backbone = backbone
        self.proposal_generator = None
        self.roi_heads = roi_heads
        self.input_format, self.vis_period = input_format, vis_period
        self.register_buffer("pixel_mean", torch.Tensor(pixel_mean).view(-1, 1, 1))
        self.register_buffer("pixel_std", torch.Tensor(pixel_std).view(-1, 1, 1))
        self.dual_stream = True
        self._side = None
        self._mean_host = [float(v) for v in pixel_mean]
        self._std_host = [float(v) for v in pixel_std]

    @property
    def device(self):
        return self.pixel_mean.device

    # ---- fused preprocess: u8 CHW -> normalised NHWC compute dtype (rcnn_multi.py:256-269)
    def _views_to_nhwc(self, imgs_u8):
        dt_ = self.backbone.compute_dtype
        cpad = 8 if dt_ == torch.bfloat16 else 4
        H, W = imgs_u8[0].shape[-2:]
        out = torch.empty(len(imgs_u8), H, W, cpad, device=self.device, dtype=dt_)
        for i, im in enumerate(imgs_u8):
            assert im.dtype == torch.uint8 and tuple(im.shape[-2:]) == (H, W)
            ops.preprocess(im.to(self.device, non_blocking=True).contiguous(), out[i], self._mean_host, self._std_host)
        return out

    def preprocess_image(self, batched_inputs):
        """API-compatible (returns 4 ImageLists of normalised f32 NCHW tensors); the training forward uses the
        fused NHWC path instead and never materialises these."""
        outs = []
        for key in ("image1", "image2", "image1_flip", "image2_flip"):
            ims = [(x[key].to(self.device).float() - self.pixel_mean) / self.pixel_std for x in batched_inputs]
            outs.append(ImageList.from_tensors(ims, self.backbone.size_divisibility))
        return tuple(outs)

    def forward(self, batched_inputs):
        assert len(batched_inputs) == 1, "now, MultiInputRCNN only support the setting -> imgs_per_gpu=1"
        if not self.training:
            return self.inference(batched_inputs)
        x = batched_inputs[0]
        for k in ("proposals1", "proposals1_flip", "proposals2", "proposals2_flip"):
            assert k in x
        # the two scales are independent until the ROI heads: run them on two HIP streams so that their ~250-workgroup
        # conv4/conv5 launches (one workgroup per CU each) share the CUs (64 KiB LDS per workgroup -> two per CU)
        proposals_list = [[x["proposals1"]], [x["proposals1_flip"]], [x["proposals2"]], [x["proposals2_flip"]]]
        gts = [[x[k]] if k in x else None for k in ("instances1", "instances1_flip", "instances2", "instances2_flip")]
        prepared = None
        if hasattr(self.roi_heads, "_prepare_inputs"):
            prepared = self.roi_heads._prepare_inputs(proposals_list, gts[0], self.device, need_grad=torch.is_grad_enabled())
        x1 = self._views_to_nhwc([x["image1"], x["image1_flip"]])
        x2 = self._views_to_nhwc([x["image2"], x["image2_flip"]])
        if self.dual_stream:
            main = torch.cuda.current_stream()
            if self._side is None:
                self._side = torch.cuda.Stream()
                # the second scale's backward runs on the side stream while .grad lives on the main one: autograd orders
                # the accumulation itself, the warning about it is noise here
                warn_off = getattr(torch.autograd.graph, "set_warn_on_accumulate_grad_stream_mismatch", None)
                if warn_off is not None:
                    warn_off(False)
            self.backbone.stage_all_weights(with_dgrad=torch.is_grad_enabled())
            self._side.wait_stream(main)
            with torch.cuda.stream(self._side):
                f2 = self.backbone.forward_nhwc(x2)
            f1 = self.backbone.forward_nhwc(x1)
            main.wait_stream(self._side)
            x2.record_stream(self._side)
            f2.record_stream(main)
        else:
            f1 = self.backbone.forward_nhwc(x1)
            f2 = self.backbone.forward_nhwc(x2)
        features1 = {"plain5": f1.permute(0, 3, 1, 2)}
        features2 = {"plain5": f2.permute(0, 3, 1, 2)}
        images_list = [None, None, None, None]       # the heads never read pixel data (roi_heads_oicrplus.py:149-188)
        if prepared is not None:
            _, detector_losses = self.roi_heads(images_list, [features1, features2], proposals_list, gts, prepared=prepared)
        else:
            _, detector_losses = self.roi_heads(images_list, [features1, features2], proposals_list, gts)
        return detector_losses          # no proposal-generator losses to merge (PrecomputedProposals); keeps LossDict.total()

    @torch.no_grad()
    def inference(self, batched_inputs, detected_instances=None, do_postprocess=True):
        assert not self.training and detected_instances is None
        x = batched_inputs[0]
        f = self.backbone.forward_nhwc(self._views_to_nhwc([x["image"]]))
        features = {"plain5": f.permute(0, 3, 1, 2)}
        proposals = [x["proposals"]]
        results, _, all_scores, all_boxes = self.roi_heads(None, features, proposals, None)
        if do_postprocess:
            image_sizes = [tuple(x["image"].shape[-2:])]
            return MultiInputRCNN._postprocess(results, batched_inputs, image_sizes)
        return results, all_scores, all_boxes

    @staticmethod
    def _postprocess(instances, batched_inputs, image_sizes):
        """rcnn_multi.py:276-291: rescale the detections to the dataset image size (`height`/`width` of the input dict)"""
        from .inference import detector_postprocess
        out = []
        for res, inp, size in zip(instances, batched_inputs, image_sizes):
            out.append({"instances": detector_postprocess(res, inp.get("height", size[0]), inp.get("width", size[1]))})
        return out
