"""CPU: the drop-in modules keep their parameter slots ([(module._parameters, name)], nets/_engine.py) instead of walking
the module tree on every call.  The slots must follow Module.parameters() order (= state_dict order, what the packers take),
see a Parameter object that was REPLACED after the first call, and the cache key must move when a value changes in place."""
import torch
from torch import nn

from shallow_wavenet_amd import config as C
from shallow_wavenet_amd.nets import cswnv_shift1 as mc
from shallow_wavenet_amd.nets import dswnv as md


def _models():
    yield mc.CSWNV(**C.tiny("laplace", seg=2, lpc=2).ctor_kwargs())
    yield md.DSWNV(**C.tiny("softmax").ctor_kwargs())


def test_slots_follow_parameters_and_state_dict_order():
    for m in _models():
        ps = m._param_list()
        assert [id(p) for p in ps] == [id(p) for p in m.parameters()]
        assert [tuple(p.shape) for p in ps] == [tuple(v.shape) for v in m.state_dict().values()]
        assert [tuple(p.shape) for p in ps] == [tuple(s) for _, s in m._cfg.param_shapes()]


def test_replaced_parameter_is_seen():
    for m in _models():
        before = m._param_list()
        new = nn.Parameter(torch.zeros_like(m.out_1.weight))
        m.out_1.weight = new
        after = m._param_list()
        assert any(p is new for p in after) and len(after) == len(before)
        assert [id(p) for p in after] == [id(p) for p in m.parameters()]


def test_key_moves_on_in_place_update():
    for m in _models():
        k0 = m._engine_key()
        assert m._engine_key() == k0
        with torch.no_grad():
            m.out_2.bias.add_(1.0)
        assert m._engine_key() != k0
