"""debug: the teacher's no-grad pass as a graph replay against plain launches, across EMA updates and interleaved student work"""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
import sos_wsod_amd  # noqa
from oracle import frcnn_oracle as FO
from test_gpu_stage3 import _model
from sos_wsod_amd.semisup import update_teacher_model
K = 20
P, PT = FO.make_params(K, tag="s3l", head_scale=14.0), FO.make_params(K, tag="s3o_teacher", head_scale=14.0)
student, teacher = _model(K, P, "s3l"), _model(K, PT, "s3l")
student.train(); teacher.train()
def data(it):
    return [{"image": torch.from_numpy(FO.make_image(160, 96, f"s3o{it}_uk0")).cuda(), "height": 160, "width": 96}]
def run(graph, it):
    os.environ["SW_S3_BACKBONE_GRAPH"] = "1" if graph else "0"
    with torch.no_grad():
        _, props, dets, _ = teacher(data(it), branch="unsup_data_weak")
    return props[0].proposal_boxes.tensor.clone(), dets[0].scores.clone(), dets[0].pred_boxes.tensor.clone()
for it in range(4):
    update_teacher_model(student, teacher, keep_rate=0.5)
    a = run(True, it)
    if os.environ.get("TWICE") and it == 0:
        a = run(True, it)
    b = run(False, it)
    print(it, "graphs:", [type(v).__name__ for v in teacher.__dict__.get("_bb_graphs", {}).values()],
          "props equal", a[0].shape == b[0].shape and torch.equal(a[0], b[0]), "det scores equal", a[1].shape == b[1].shape and torch.equal(a[1], b[1]),
          "max score diff", float((a[1] - b[1]).abs().max()) if a[1].shape == b[1].shape and a[1].numel() else None)
