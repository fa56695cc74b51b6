"""CPU: the happens-before check of the following scheme's hand-off protocol (tests/protocol_model.py) on the host's own
task lists -- every read of a shared object ordered after its write and before the next one, by explicit hand-offs only --
and the proof that the check SEES the two holes round 6 closed (it fails when either is put back)."""
import pytest

from protocol_model import Model, Model01


@pytest.mark.parametrize("P", [1, 2, 3, 5, 8, 10])
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


@pytest.mark.parametrize("scheme,P", [(1, 1), (1, 2), (1, 4), (1, 7), (1, 10), (0, 1), (0, 2), (0, 4), (0, 7), (0, 10)])
def test_schemes_0_and_1_are_ordered_by_hand_offs_alone(scheme, P):
    """... and without the in-order publication too: in these schemes every publisher of potrf_done / rows_done depends, through
    its data, on its predecessor's publication (the hole of round 6 was specific to the second level of following)"""
    assert Model01(P, scheme).races() == []
    assert Model01(P, scheme, inorder=False).races() == []


@pytest.mark.parametrize("scheme,P,drop,where,seen", [
    (0, 8, "rows_done", "DIAG", "'wt'"),          # the wait ahead of potrf in scheme 0: it keeps the W tile of block q - 3 for its readers
    (1, 6, "off1_ready", "DIAG", "'tile', 0, 1"),  # the fused diagonal task would solve a tile that is not updated yet
    (0, 6, "potrf_done", "OFF", "'rv', 0"),        # a strip solve without its wait for the factorisation
])
def test_the_model_notices_a_wait_that_is_taken_away(monkeypatch, scheme, P, drop, where, seen):
    """the clean results above are not vacuous: remove one wait of the kernel from the model and the pair it ordered shows up"""
    import protocol_model as pm
    orig = pm.Model.Seq.wait

    def wait(self, flag, target, counter=False):
        if flag == drop and where in self.label:
            return
        return orig(self, flag, target, counter)
    monkeypatch.setattr(pm.Model.Seq, "wait", wait)
    races = Model01(P, scheme).races()
    assert races and any(seen in r for r in races), races[:3]


@pytest.mark.parametrize("P,Mt", [(1, 1), (3, 2), (6, 3), (8, 2)])
def test_the_augmented_launch_of_predict_is_ordered_by_hand_offs_alone(P, Mt):
    """[B | Cx^T] factored by one launch with the prediction columns as extra column tiles and Sigma's tiles as Schur tasks"""
    assert Model(P, Mt=Mt, Ms=Mt).races() == []
    assert Model(P, Mt=Mt, Ms=Mt, inorder=False).races() != [] or P < 3
