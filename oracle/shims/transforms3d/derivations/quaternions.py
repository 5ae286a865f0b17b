from .._impl import qmult as _qmult


def qmult(q1, q2):
    """(the derivations variant returns a tuple)"""
    return tuple(_qmult(q1, q2))
