"""WSDDN predictor: parameter holder with the reference's names/initialisation
(uwsod/projects/WSL/wsl/modeling/roi_heads/fast_rcnn_wsddn.py:432-589: `cls`, `det` Linear(4096->K),
xavier-uniform weights, zero bias).  The arithmetic (softmax over classes x softmax over proposals,
image-level BCE :340-375) is the fused kernel sw_wsddn_mil, driven from OICRPlusHeads."""
import torch
import torch.nn as nn

from .box_head import _Linear


class WSDDNOutputLayers(nn.Module):
    def __init__(self, input_shape, *, box2box_transform=None, num_classes, cls_agnostic_bbox_reg=False,
                 smooth_l1_beta=0.0, test_score_thresh=0.0, test_nms_thresh=0.5, test_topk_per_image=100,
                 box_reg_loss_type="smooth_l1", loss_weight=1.0, mean_loss=True, **unused):
        super().__init__()
        d = input_shape if isinstance(input_shape, int) else input_shape.channels * (input_shape.width or 1) * (input_shape.height or 1)
        assert mean_loss, "WSL.MEAN_LOSS True is the configured path (voc07_oicr_plus.yaml)"
        self.num_classes = num_classes
        self.cls = _Linear(d, num_classes)
        self.det = _Linear(d, num_classes)
        nn.init.xavier_uniform_(self.cls.weight)          # fast_rcnn_wsddn.py:495-498
        nn.init.xavier_uniform_(self.det.weight)
        nn.init.constant_(self.cls.bias, 0)
        nn.init.constant_(self.det.bias, 0)
        if isinstance(loss_weight, float):
            loss_weight = {"loss_cls": loss_weight, "loss_box_reg": loss_weight}
        self.loss_weight = loss_weight

    # ---- stand-alone call API (predictor_api.py); OICRPlusHeads' training path runs the same kernels fused over all views
    compute_dtype = torch.float32

    def forward(self, x, proposals=None, context=False):
        """fast_rcnn_wsddn.py:542-589 -> (scores (N, K), zero proposal_deltas (N, 4K))"""
        assert not context, "contextlocnet is not on the OICR+ path"
        from .predictor_api import wsddn_forward
        return wsddn_forward(self, x, proposals, self.compute_dtype)

    def losses(self, predictions, proposals, gt_classes_img_oh):
        """fast_rcnn_wsddn.py:658-681 -> {"loss_cls": ...}"""
        from .predictor_api import wsddn_losses
        return wsddn_losses(self, predictions, proposals, gt_classes_img_oh)

    def predict_probs(self, predictions, proposals):
        """fast_rcnn_wsddn.py:683-697: the scores, split per image"""
        scores, _ = predictions
        return scores.split([len(p) for p in proposals], dim=0)

    @classmethod
    def from_config(cls, cfg, input_shape):
        return dict(input_shape=input_shape, num_classes=cfg.MODEL.ROI_HEADS.NUM_CLASSES,
                    cls_agnostic_bbox_reg=cfg.MODEL.ROI_BOX_HEAD.CLS_AGNOSTIC_BBOX_REG,
                    smooth_l1_beta=cfg.MODEL.ROI_BOX_HEAD.SMOOTH_L1_BETA,
                    test_score_thresh=cfg.MODEL.ROI_HEADS.SCORE_THRESH_TEST,
                    test_nms_thresh=cfg.MODEL.ROI_HEADS.NMS_THRESH_TEST,
                    test_topk_per_image=cfg.TEST.DETECTIONS_PER_IMAGE,
                    box_reg_loss_type=cfg.MODEL.ROI_BOX_HEAD.BBOX_REG_LOSS_TYPE,
                    loss_weight={"loss_box_reg": cfg.MODEL.ROI_BOX_HEAD.BBOX_REG_LOSS_WEIGHT},
                    mean_loss=cfg.WSL.MEAN_LOSS)
