"""Known-answer cases for the greedy NMS core (restates torchvision.ops.nms; "parity unpinned":
the third-party source is absent, so these hand-computed cases are the pin)."""
import torch

from oracle.postproc import greedy_nms


def test_chain_suppression_is_greedy_not_transitive():
    # A overlaps B (IoU .6), B overlaps C (.6), A-C (.333).  A kept -> B dies -> C survives.
    b = torch.tensor([[0., 0., 10., 10.], [2.5, 0., 12.5, 10.], [5., 0., 15., 10.]])
    s = torch.tensor([0.9, 0.8, 0.7])
    assert greedy_nms(b, s, 0.5).tolist() == [0, 2]
    assert greedy_nms(b, s, 0.3).tolist() == [0]          # .333 > .3 kills C too
    assert greedy_nms(b, s, 0.7).tolist() == [0, 1, 2]


def test_threshold_is_strict():
    # IoU exactly 0.5: inter 50, union 100 (two 10x7.5 boxes sharing 10x5)  -> not suppressed at thr 0.5
    b = torch.tensor([[0., 0., 10., 7.5], [0., 2.5, 10., 10.]])
    s = torch.tensor([0.6, 0.5])
    assert greedy_nms(b, s, 0.5).tolist() == [0, 1]
    assert greedy_nms(b, s, 0.4999).tolist() == [0]


def test_output_is_in_score_order_and_disjoint_boxes_all_kept():
    b = torch.tensor([[0., 0., 1., 1.], [5., 5., 6., 6.], [10., 10., 11., 11.], [20., 0., 21., 1.]])
    s = torch.tensor([0.1, 0.9, 0.5, 0.7])
    assert greedy_nms(b, s, 0.45).tolist() == [1, 3, 2, 0]


def test_degenerate_zero_area_boxes_never_suppress():
    # 0/0 IoU is NaN -> comparison false -> kept (torchvision semantics: no eps)
    b = torch.tensor([[1., 1., 1., 1.], [1., 1., 1., 1.], [0., 0., 4., 4.]])
    s = torch.tensor([0.9, 0.8, 0.7])
    assert greedy_nms(b, s, 0.45).tolist() == [0, 1, 2]


def test_contained_box():
    b = torch.tensor([[0., 0., 10., 10.], [2., 2., 8., 8.], [0., 0., 10., 10.2]])
    s = torch.tensor([0.5, 0.9, 0.7])       # order: 1, 2, 0 ; IoU(1,2)=36/102, IoU(2,0)=100/102
    assert greedy_nms(b, s, 0.45).tolist() == [1, 2]
    assert greedy_nms(b, s, 0.3).tolist() == [1]


def test_tie_break_lower_index_first():
    b = torch.tensor([[0., 0., 10., 10.], [0., 0., 10., 10.]])
    s = torch.tensor([0.5, 0.5])
    assert greedy_nms(b, s, 0.45).tolist() == [0]


def test_empty():
    assert greedy_nms(torch.zeros(0, 4), torch.zeros(0), 0.5).numel() == 0
