"""ORACLE-ONLY TOOL: the four dgl.function builtins the reference model uses
(src/model/rigid_docking_model.py:204-205, 274-283)."""


class _USubV:
    def __init__(self, lhs, rhs, out):
        self.lhs, self.rhs, self.out = lhs, rhs, out


class _CopyEdge:
    def __init__(self, field, out):
        self.field, self.out = field, out


class _Mean:
    def __init__(self, msg, out):
        self.msg, self.out = msg, out


def u_sub_v(lhs, rhs, out):
    return _USubV(lhs, rhs, out)


def copy_edge(field, out):
    return _CopyEdge(field, out)


copy_e = copy_edge


def mean(msg, out):
    return _Mean(msg, out)
