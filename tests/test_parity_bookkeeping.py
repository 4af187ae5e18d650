"""The parity bookkeeping itself (CPU): the table of documented exceptions pins what it claims to pin."""
import re

import parity_exceptions as pe


def test_no_exception_bound_is_wider_than_four_times_its_measurement():
    """Verdict r5 item 8: every entry of tests/parity_exceptions.py states the value measured for it (`measured a / b / c at ... absolute`,
    relative measure first); the bound held is at most 4 x the largest of them and never below it."""
    for pat, bound, why in pe.TABLE:
        head = why.split(";")[0].split(":")[0]
        m = re.match(r"measured ([0-9.e\- /]+?) at ", head)
        assert m, (pat, head)
        measured = [float(v) for v in m.group(1).split("/")]
        assert max(measured) <= bound <= 4.0 * max(measured), (pat, bound, measured)


def test_exception_lookup_matches_whole_labels_only():
    assert pe.lookup("compute_sdf_alpha alpha cos_anneal=0.5")[0] == 1e-2
    assert pe.lookup("module_tensosdf:195.1")[0] == 2e-2
    assert pe.lookup("module_tensosdf:195.10") is None and pe.lookup(None) is None and pe.lookup("colors") is None
