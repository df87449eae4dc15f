"""Abaqus .inp reader with the call surface of the reference's `parse_inp.InpParser` (parse_inp.py:8-109;
examples/torque_arm/arm.py:19-21 is its user): `InpParser(path).parse() -> (conn, X, groups)`.

Same conventions: node and element ids are 1-based in the file and must be contiguous inside a section (the
section is sorted by id, so the file order does not matter); connectivity and node sets come out 0-based;
`conn` is a dict keyed by the Abaqus element type (only CPS3, C3D8R and C3D10 sections are kept, a later section
of the same type replaces an earlier one); `X` has the three coordinates of the file.  Written as a single pass
over the sections with numpy doing the number parsing."""
import os
import re

import numpy as np

from . import utils

SUPPORTED_ELEMENT = {"CPS3": 3, "C3D8R": 8, "C3D10": 10}  # type -> nodes per element


class InpParser:
    def __init__(self, inp_name):
        self.inp_name = inp_name
        self.conn = None
        self.X = None
        self.groups = None

    @staticmethod
    def _sections(path):
        """Yield (keyword, {option: value}, [data lines]) for every *KEYWORD section ('**' lines are comments)."""
        keyword, options, lines = None, {}, []
        with open(path, "r") as fh:
            for raw in fh:
                line = raw.strip()
                if not line or line.startswith("**"):
                    continue
                if line.startswith("*"):
                    if keyword is not None:
                        yield keyword, options, lines
                    head = [p.strip() for p in line[1:].split(",")]
                    keyword = head[0].lower()
                    options = {}
                    for p in head[1:]:
                        m = re.fullmatch(r"(\w+)\s*=\s*(\S+)", p)
                        if m:
                            options[m.group(1).lower()] = m.group(2)
                    lines = []
                elif keyword is not None:
                    lines.append(line)
        if keyword is not None:
            yield keyword, options, lines

    @staticmethod
    def _table(lines, dtype):
        """Rows of comma-separated numbers (trailing commas allowed) as a 2-D array."""
        return np.array([[dtype(v) for v in ln.strip(", ").split(",")] for ln in lines])

    @staticmethod
    def _by_id(table, what):
        ids = table[:, 0].astype(np.int64)
        if not (np.unique(ids).size == ids.size == ids.max() - ids.min() + 1):
            raise AssertionError(f"{what} ids must be unique and contiguous inside a section")
        return table[np.argsort(ids, kind="stable"), 1:]

    def parse(self):
        nodes, conn, groups = [], {}, {}
        for keyword, opt, lines in self._sections(self.inp_name):
            if keyword == "node":
                nodes.append(self._by_id(self._table(lines, float), "node"))
            elif keyword == "element" and opt.get("type") in SUPPORTED_ELEMENT:
                conn[opt["type"]] = self._by_id(self._table(lines, int), "element").astype(np.int64) - 1
            elif keyword == "nset":
                flat = [int(v) for ln in lines for v in ln.strip(", ").split(",")]
                groups[opt.get("nset")] = np.array(flat, dtype=np.int64) - 1
        if not nodes:
            raise ValueError(f"{self.inp_name}: no *NODE section")
        if len(nodes) > 1:
            print("[Warning] Multiple *Node sections detected")
        self.conn, self.X, self.groups = conn, np.concatenate(nodes, axis=0), groups
        return self.conn, self.X, self.groups

    def to_vtk(self, nodal_sol={}, vtk_name=None):
        """ref: parse_inp.py:84-108."""
        if self.conn is None:
            self.parse()
        if vtk_name is None:
            vtk_name = os.path.splitext(self.inp_name)[0] + ".vtk"
        utils.to_vtk(self.conn, self.X, nodal_sol, vtk_name)
