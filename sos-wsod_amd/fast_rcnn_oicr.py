"""OICR refinement predictor: parameter holder with the reference's names/initialisation
(uwsod/projects/WSL/wsl/modeling/roi_heads/fast_rcnn_oicr.py:408-528: `cls_score` Linear(4096->K+1) N(0,0.01),
`bbox_pred` Linear(4096->4K) N(0,0.001), zero bias).  Losses (:157-352) are the fused kernel
sw_oicr_refine_loss; inference utilities (:46-148, :674-735) live in OICRPlusHeads._forward_box_test."""
import torch
import torch.nn as nn

from .box_head import _Linear


class OICROutputLayers(nn.Module):
    def __init__(self, input_shape, *, box2box_transform=None, num_classes, cls_agnostic_bbox_reg=False,
                 smooth_l1_beta=0.0, test_score_thresh=0.0, test_nms_thresh=0.5, test_topk_per_image=100,
                 box_reg_loss_type="smooth_l1", loss_weight=1.0, mean_loss=True, refine_k=0, refine_reg=None, **unused):
        super().__init__()
        d = input_shape if isinstance(input_shape, int) else input_shape.channels * (input_shape.width or 1) * (input_shape.height or 1)
        assert not cls_agnostic_bbox_reg, "CLS_AGNOSTIC_BBOX_REG False on this path"
        assert smooth_l1_beta == 0.0 and box_reg_loss_type == "smooth_l1", "SMOOTH_L1_BETA 0.0 => L1 (defaults.py:298)"
        refine_reg = refine_reg if refine_reg is not None else [True] * (refine_k + 1)
        assert refine_reg[refine_k], "WSL.REFINE_REG all True on this path"
        self.num_classes, self.refine_k = num_classes, refine_k
        self.bbox_reg_weights = tuple(getattr(box2box_transform, "weights", (10.0, 10.0, 5.0, 5.0)))     # defaults.py:296
        self.cls_score = _Linear(d, num_classes + 1)
        self.bbox_pred = _Linear(d, 4 * num_classes)
        nn.init.normal_(self.cls_score.weight, std=0.01)      # fast_rcnn_oicr.py:474-478
        nn.init.normal_(self.bbox_pred.weight, std=0.001)
        nn.init.constant_(self.cls_score.bias, 0)
        nn.init.constant_(self.bbox_pred.bias, 0)
        self.test_score_thresh, self.test_nms_thresh, self.test_topk_per_image = test_score_thresh, test_nms_thresh, test_topk_per_image
        if isinstance(loss_weight, float):
            loss_weight = {"loss_cls": loss_weight, "loss_box_reg": loss_weight}
        self.loss_weight = loss_weight

    # ---- stand-alone call API (predictor_api.py); OICRPlusHeads' training path runs the same kernels fused over all views
    compute_dtype = torch.float32

    def forward(self, x):
        """fast_rcnn_oicr.py:504-528 -> (scores (N, K+1), proposal_deltas (N, 4K))"""
        from .predictor_api import oicr_forward
        return oicr_forward(self, x, self.compute_dtype)

    def losses(self, predictions, proposals):
        """fast_rcnn_oicr.py:530-554 -> {"loss_cls", "loss_box_reg"}; proposals: proposal_boxes, gt_boxes, gt_classes, gt_weights"""
        from .predictor_api import oicr_losses
        return oicr_losses(self, predictions, proposals)

    def predict_probs(self, predictions, proposals):
        """fast_rcnn_oicr.py:702-716"""
        from .predictor_api import oicr_predict_probs
        return oicr_predict_probs(self, predictions, proposals)

    def inference(self, predictions, proposals):
        """fast_rcnn_oicr.py:584-614"""
        from .predictor_api import oicr_inference
        return oicr_inference(self, predictions, proposals)

    @classmethod
    def from_config(cls, cfg, input_shape, k):
        return dict(input_shape=input_shape, num_classes=cfg.MODEL.ROI_HEADS.NUM_CLASSES,
                    cls_agnostic_bbox_reg=cfg.MODEL.ROI_BOX_HEAD.CLS_AGNOSTIC_BBOX_REG,
                    smooth_l1_beta=cfg.MODEL.ROI_BOX_HEAD.SMOOTH_L1_BETA,
                    test_score_thresh=cfg.MODEL.ROI_HEADS.SCORE_THRESH_TEST,
                    test_nms_thresh=cfg.MODEL.ROI_HEADS.NMS_THRESH_TEST,
                    test_topk_per_image=cfg.TEST.DETECTIONS_PER_IMAGE,
                    box_reg_loss_type=cfg.MODEL.ROI_BOX_HEAD.BBOX_REG_LOSS_TYPE,
                    loss_weight={"loss_box_reg": cfg.MODEL.ROI_BOX_HEAD.BBOX_REG_LOSS_WEIGHT},
                    mean_loss=cfg.WSL.MEAN_LOSS, refine_k=k, refine_reg=cfg.WSL.REFINE_REG)
