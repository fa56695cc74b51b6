"""CPU: the happens-before check of the following scheme's hand-off protocol (tests/protocol_model.py) on the host's own
task lists -- every read of a shared object ordered after its write and before the next one, by explicit hand-offs only --
and the proof that the check SEES the two holes round 6 closed (it fails when either is put back)."""
import pytest

from protocol_model import Model


@pytest.mark.parametrize("P", [1, 2, 3, 5, 8, 11])
def test_following_scheme_is_ordered_by_hand_offs_alone(P):
    m = Model(P)
    assert m.races() == []


def test_the_model_sees_the_accumulator_chain_of_rounds_1_to_5():
    """one record per matrix, read-modify-written by every diagonal task: diagonal task q + 1 follows the STEPS of task q,
    nothing orders the two updates (the lost block row of profiles/r6_acc_forensics.txt)"""
    races = Model(6, acc_chain=True).races()
    assert races and all("('acc',)" in r for r in races), races[:3]
    assert any("DIAG(2)" in r or "DIAG(3)" in r for r in races)


def test_the_model_sees_the_unordered_progress_words_of_rounds_3_to_5():
    """potrf_done / rows_done published without waiting for the predecessor's value: a late diagonal task lets its successor's
    larger value satisfy the strip solves that wait for its own z"""
    races = Model(6, inorder=False).races()
    assert races, "publication order should matter"
    assert any("rv" in r for r in races), races[:5]


def test_the_model_sees_a_missing_right_hand_side_wait():
    """round 3's own find (dag_pss, rv_wait): without the wait for the strip solve of the row above, two tasks update one block
    of the right-hand side with nothing between them"""
    races = Model(5, rv_wait=False).races()
    assert races and any("('rv'," in r for r in races)
