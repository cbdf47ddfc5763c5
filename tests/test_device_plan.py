"""The device an encoder-pool worker or a decoder handle is dealt on a node with several GPUs (cfhd_core.h unit_device, host logic):
round robin like the reference's pool deals frames to its encoder threads (EncoderSDK/EncoderPool.cpp:281-291), everything on the
process's own device when a launcher pinned one (one process per GPU: bench.py under torch.distributed.run), an explicit list otherwise."""
import ctypes
from cfhd_testlib import hooks


def _dev(i, n, pinned=None, lst=None):
    L = hooks()
    L.cfhd_amd_unit_device.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_char_p, ctypes.c_char_p]
    L.cfhd_amd_unit_device.restype = ctypes.c_int
    return L.cfhd_amd_unit_device(i, n, pinned.encode() if pinned is not None else None, lst.encode() if lst is not None else None)


def test_pool_workers_spread_round_robin_over_the_gpus_of_a_node():
    assert [_dev(i, 8) for i in range(16)] == [0, 1, 2, 3, 4, 5, 6, 7] * 2
    assert [_dev(i, 1) for i in range(4)] == [0, 0, 0, 0]
    assert [_dev(i, 3) for i in range(7)] == [0, 1, 2, 0, 1, 2, 0]
    assert _dev(5, 0) == 0                                   # (no device count yet: device 0)


def test_one_process_per_gpu_keeps_every_unit_on_its_own_device():
    # CFHD_AMD_DEVICE / LOCAL_RANK set by the launcher: -1 = the process default, whatever the node holds
    assert [_dev(i, 8, pinned="3") for i in range(5)] == [-1] * 5
    assert _dev(2, 8, pinned="") == 2                        # an empty variable pins nothing


def test_an_explicit_device_list_wins_and_may_repeat_devices():
    assert [_dev(i, 8, lst="4,5") for i in range(5)] == [4, 5, 4, 5, 4]
    assert [_dev(i, 8, pinned="1", lst="0,0,2") for i in range(4)] == [0, 0, 2, 0]      # the list beats the pin
    assert [_dev(i, 2, lst="0, 3 ,1") for i in range(3)] == [0, 1, 1]                   # entries are taken modulo the device count
    assert [_dev(i, 4, lst="x") for i in range(3)] == [0, 1, 2]                         # nothing usable in the list: round robin
