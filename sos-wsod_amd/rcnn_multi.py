"""MultiInputRCNN meta-architecture behind the reference's interface
(uwsod/detectron2/modeling/meta_arch/rcnn_multi.py:23-291; build_model meta_arch/build.py:15-23).

`model(batched_inputs) -> dict of 9 losses` in training, `list[{"instances": Instances}]` in eval; same
input dict keys (image{1,2}{,_flip} u8 CHW BGR, proposals*, instances*); the reference's 1-image-per-GPU assert (:148) is lifted.
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
        self._mean_host = [float(v) for v in pixel_mean]
        self._std_host = [float(v) for v in pixel_std]

    @property
    def device(self):
        return self.pixel_mean.device

    def prepare_for_training(self):
        """Everything that re-homes parameter storage or builds persistent compute-dtype weight copies, done ONCE and before a
        DistributedDataParallel wrapper inspects the parameters: the 10 predictor weights become row slices of one flat
        master (roi_heads._flatten_head_params), conv / fc weights get their kernel-layout copies."""
        dev = self.device
        if dev.type != "cuda":
            return
        hd = self.roi_heads
        if hasattr(hd, "_flatten_head_params") and not hd._head_flat_ok(hd._flat_params(), dev):
            hd._flatten_head_params(dev)
        if hasattr(self.backbone, "stage_all_weights"):
            self.backbone.stage_all_weights(with_dgrad=True)

    # ---- fused preprocess: u8 CHW -> normalised NHWC compute dtype (rcnn_multi.py:256-269)
    def _views_to_nhwc(self, imgs_u8):
        dt_ = self.backbone.compute_dtype
        cpad = 8 if dt_ == torch.bfloat16 else 4
        H, W = imgs_u8[0].shape[-2:]
        out = torch.empty(len(imgs_u8), H, W, cpad, device=self.device, dtype=dt_)
        ims = []
        for im in imgs_u8:
            assert im.dtype == torch.uint8 and tuple(im.shape[-2:]) == (H, W)
            ims.append(im.to(self.device, non_blocking=True).contiguous())
        ops.preprocess_multi(ims, out, self._mean_host, self._std_host)       # the whole view batch in one launch
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
        """rcnn_multi.py:125-208.  The reference asserts ONE image per GPU (:148); here a batch of B images is B times the
        same computation (every image keeps its own view sizes, proposal count, MIL softmax, mining and NMS) on stacked
        rows, the losses averaged over the images — exactly what DDP forms over B ranks of one image each."""
        if not self.training:
            assert len(batched_inputs) == 1, "inference runs one image per call (rcnn_multi.py:148)"
            return self.inference(batched_inputs)
        for x in batched_inputs:
            for k in ("proposals1", "proposals1_flip", "proposals2", "proposals2_flip"):
                assert k in x
        proposals_list = [[x[k] for x in batched_inputs] for k in ("proposals1", "proposals1_flip", "proposals2", "proposals2_flip")]
        gts = [[x[k] for x in batched_inputs] if all(k in x for x in batched_inputs) else None
               for k in ("instances1", "instances1_flip", "instances2", "instances2_flip")]
        prepared = None
        if hasattr(self.roi_heads, "_prepare_inputs"):
            prepared = self.roi_heads._prepare_inputs(proposals_list, gts[0], self.device, need_grad=torch.is_grad_enabled())
        # one NHWC batch of 2 (view, flipped view) per scale and image; the backbone runs all of them as ONE autograd node
        # whose batches alternate between two HIP streams (backbone_vgg._VGGFunction)
        xs = []
        for x in batched_inputs:
            xs.append(self._views_to_nhwc([x["image1"], x["image1_flip"]]))
            xs.append(self._views_to_nhwc([x["image2"], x["image2_flip"]]))
        self.backbone.dual_stream = self.dual_stream
        fs = self.backbone.forward_views(xs)
        features = [{"plain5": f.permute(0, 3, 1, 2)} for f in fs]            # NCHW views, the reference's feature format
        images_list = [None, None, None, None]       # the heads never read pixel data (roi_heads_oicrplus.py:149-188)
        if prepared is not None:
            _, detector_losses = self.roi_heads(images_list, features, proposals_list, gts, prepared=prepared)
        else:
            _, detector_losses = self.roi_heads(images_list, features, proposals_list, gts)
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

    @torch.no_grad()
    def view_scores(self, views):
        """per-view class scores and decoded boxes of several views of ONE size in one pass (test-time augmentation: a view and its
        flip): backbone on the batch, heads on stacked rows -> [(scores (R, K+1), boxes (R, 4K))] in view coordinates"""
        from .inference import oicr_view_scores
        assert not self.training and len({tuple(v["image"].shape[-2:]) for v in views}) == 1
        f = self.backbone.forward_nhwc(self._views_to_nhwc([v["image"] for v in views]))
        return oicr_view_scores(self.roi_heads, f, [v["proposals"] for v in views])

    @staticmethod
    def _postprocess(instances, batched_inputs, image_sizes):
        """rcnn_multi.py:276-291: rescale the detections to the dataset image size (`height`/`width` of the input dict)"""
        from .inference import detector_postprocess
        out = []
        for res, inp, size in zip(instances, batched_inputs, image_sizes):
            out.append({"instances": detector_postprocess(res, inp.get("height", size[0]), inp.get("width", size[1]))})
        return out
