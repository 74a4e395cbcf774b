"""Inspection of a captured train step (hipGraph): node-type histogram through the HIP runtime's own graph API.

A memset node inside a replayed graph is executed wrongly by this ROCm release from the second replay on (DESIGN.md section 5,
tools/probes/graph_memset_probe.py): the library issues none (zero fills are kernels), and ``assert_no_memset_nodes`` checks the
CAPTURED graph itself -- torch-side ops and RCCL's nodes included -- instead of grepping sources.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict

import torch

NODE_TYPES = ("kernel", "memcpy", "memset", "host", "graph", "empty", "wait_event", "event_record", "ext_sem_signal", "ext_sem_wait",
              "mem_alloc", "mem_free", "memcpy_from_symbol", "memcpy_to_symbol", "batch_mem_op")      # hipGraphNodeType, hip_runtime_api.h
_hip = None


def _runtime():
    global _hip
    if _hip is None:
        err = None
        for name in ("libamdhip64.so.7", "libamdhip64.so", "libamdhip64.so.6"):
            try:
                _hip = C.CDLL(name)
                break
            except OSError as e:  # noqa: PERF203
                err = e
        if _hip is None:
            raise RuntimeError(f"HIP runtime not loadable for graph inspection: {err}")
        _hip.hipGraphGetNodes.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
        _hip.hipGraphGetNodes.restype = C.c_int
        _hip.hipGraphNodeGetType.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
        _hip.hipGraphNodeGetType.restype = C.c_int
        _hip.hipGraphChildGraphNodeGetGraph.argtypes = [C.c_void_p, C.POINTER(C.c_void_p)]
        _hip.hipGraphChildGraphNodeGetGraph.restype = C.c_int
    return _hip


def _count(graph_ptr: int, hist: Dict[str, int]) -> None:
    hip = _runtime()
    n = C.c_size_t(0)
    if hip.hipGraphGetNodes(C.c_void_p(graph_ptr), None, C.byref(n)) != 0:
        raise RuntimeError("hipGraphGetNodes failed")
    if n.value == 0:
        return
    nodes = (C.c_void_p * n.value)()
    if hip.hipGraphGetNodes(C.c_void_p(graph_ptr), nodes, C.byref(n)) != 0:
        raise RuntimeError("hipGraphGetNodes failed")
    for i in range(n.value):
        t = C.c_int(-1)
        if hip.hipGraphNodeGetType(nodes[i], C.byref(t)) != 0:
            raise RuntimeError("hipGraphNodeGetType failed")
        name = NODE_TYPES[t.value] if 0 <= t.value < len(NODE_TYPES) else f"type{t.value}"
        hist[name] = hist.get(name, 0) + 1
        if name == "graph":                                  # (RCCL may capture child graphs: their nodes count too)
            child = C.c_void_p(0)
            if hip.hipGraphChildGraphNodeGetGraph(nodes[i], C.byref(child)) == 0 and child.value:
                _count(child.value, hist)


def node_histogram(graph: torch.cuda.CUDAGraph) -> Dict[str, int]:
    """{node type: count} of a graph captured with ``torch.cuda.CUDAGraph(keep_graph=True)`` (the raw hipGraph_t must still exist;
    call before or after ``instantiate``)."""
    hist: Dict[str, int] = {}
    _count(int(graph.raw_cuda_graph()), hist)
    return hist


def assert_no_memset_nodes(graph: torch.cuda.CUDAGraph) -> Dict[str, int]:
    hist = node_histogram(graph)
    if hist.get("memset", 0):
        raise RuntimeError(f"the captured step holds {hist['memset']} memset node(s): {hist} -- a replayed memset node is executed wrongly "
                           "on this ROCm release (DESIGN.md 5); replace the fill by a kernel (m1_zero / torch fill_)")
    return hist
