"""ctypes mirrors of the POD structs in include/hrgym.h and include/hrgym_state.h.

The mirrors are generated from the headers themselves (one source of truth); `sizeof` is cross-checked
against the compiled library (`hrg_state_bytes`) when it is loaded.
"""
import ctypes
import os
import re

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_INCLUDE = os.path.join(_ROOT, "include")

_CTYPES = {
    "double": ctypes.c_double,
    "float": ctypes.c_float,
    "int32_t": ctypes.c_int32,
    "uint32_t": ctypes.c_uint32,
    "int64_t": ctypes.c_int64,
    "uint64_t": ctypes.c_uint64,
    "uint8_t": ctypes.c_uint8,
    "const double*": ctypes.POINTER(ctypes.c_double),
}


def _strip_comments(src):
    return re.sub(r"/\*.*?\*/", "", src, flags=re.S)


def _parse(paths):
    consts, structs = {}, {}
    for path in paths:
        src = _strip_comments(open(path).read())
        for m in re.finditer(r"#define\s+(HRG_\w+)\s+(.+)", src):
            expr = m.group(2).strip()
            try:
                consts[m.group(1)] = int(eval(expr, {}, consts))
            except Exception:
                pass
        for m in re.finditer(r"enum\s*\{(.*?)\}", src, flags=re.S):
            val = -1
            for item in m.group(1).split(","):
                item = item.strip()
                if not item:
                    continue
                if "=" in item:
                    name, v = [x.strip() for x in item.split("=")]
                    val = int(eval(v, {}, consts))
                else:
                    name, val = item, val + 1
                consts[name] = val
        for m in re.finditer(r"typedef struct (\w+) \{(.*?)\}\s*(\w+);", src, flags=re.S):
            fields = []
            for decl in m.group(2).split(";"):
                decl = " ".join(decl.split())
                if not decl:
                    continue
                mm = re.match(r"(const double\*|\w+)\s+(.*)", decl)
                tname, rest = mm.group(1), mm.group(2)
                base = structs[tname] if tname in structs else _CTYPES[tname]
                for var in rest.split(","):
                    var = var.strip()
                    name = re.match(r"\w+", var).group(0)
                    dims = [int(eval(d, {}, consts)) for d in re.findall(r"\[([^\]]+)\]", var)]
                    t = base
                    for d in reversed(dims):
                        t = t * d
                    fields.append((name, t))
            structs[m.group(3)] = type(m.group(3), (ctypes.Structure,), {"_fields_": fields})
    return consts, structs


CONST, _STRUCTS = _parse([os.path.join(_INCLUDE, "hrgym.h"), os.path.join(_INCLUDE, "hrgym_state.h")])
ModelDesc = _STRUCTS["hrg_model_desc"]
ClipTable = _STRUCTS["hrg_clip_table"]
LTT = _STRUCTS["hrg_ltt"]
Path = _STRUCTS["hrg_path"]
EnvState = _STRUCTS["hrg_env_state"]
BoxState = _STRUCTS["hrg_box_state"]
StackState = _STRUCTS["hrg_stack_state"]
HammerState = _STRUCTS["hrg_hammer_state"]


def struct_to_dict(s):
    """Recursively convert a ctypes struct/array to python lists/dicts (for parity comparisons)."""
    if isinstance(s, ctypes.Structure):
        return {n: struct_to_dict(getattr(s, n)) for n, _ in s._fields_}
    if isinstance(s, ctypes.Array):
        return [struct_to_dict(x) for x in s]
    return s
