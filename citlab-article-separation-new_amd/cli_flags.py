"""Command-line grammar of the reference's flag system (``python_util/basic/flags.py``), restated on argparse:

  * ``@path/to/config`` files whose lines hold ``--flag value ...``, ``=`` separators and ``#`` comments (flags.py:10-28);
  * dict flags ``--input_params k=v k2=[a,b]`` with automatic typing: true/t/false/f -> bool, numbers -> int when integral
    else float, ``[..]`` -> list of typed elements, everything else stays a string (flags.py:228-285);
  * ``update_params``: unknown keys are reported with ``logging.critical`` but still merged (flags.py:303-333).

Pinned by golden vectors captured from the imported reference (tests/golden/make_flags_golden.py).
"""
import argparse
import logging


class LineArgumentParser(argparse.ArgumentParser):
    def convert_arg_line_to_args(self, arg_line):
        args = arg_line.split()
        for i, arg in enumerate(args):
            if arg == "#":
                return args[:i]
            if arg == "=":
                args.remove("=")
        return args


def _typed(s):
    low = s.lower()
    if low in ("true", "t"):
        return True
    if low in ("false", "f"):
        return False
    try:
        f = float(s)
    except ValueError:
        return s
    i = int(f)
    return i if i == f else f


def parse_key_value(kv_list, into=None):
    """The body of StoreDictKeyPair.__call__ (flags.py:252-285)."""
    out = {} if into is None else into
    for kv in kv_list:
        parts = kv.split("=")
        if len(parts) != 2:
            continue
        key, val = parts
        s = val.strip()
        typed = _typed(val)
        if isinstance(typed, str) and len(s) >= 1 and s[0] == "[" and s[-1] == "]":
            elems = [e.strip() for e in s[1:-1].split(",")]
            typed = [_typed(e) for e in elems if e != ""]
        out[key] = typed
    return out


class StoreDictKeyPair(argparse.Action):
    def __call__(self, parser, namespace, values, option_string=None):
        if not getattr(namespace, self.dest):
            setattr(namespace, self.dest, {})
        parse_key_value(values, getattr(namespace, self.dest))


def define_dict(parser, name, default, doc=""):
    parser.add_argument("--" + name, action=StoreDictKeyPair, default=default, nargs="*", metavar="KEY=VAL", help=doc)


def str2bool(v):
    if isinstance(v, bool):
        return v
    return str(v).lower() in ("true", "t", "1")


def update_params(class_params, flag_params, name=""):
    for k in flag_params:
        if k not in class_params:
            logging.critical("Given {0}_params-key '{1}' is not used by {0}-class!".format(name, k))
    class_params.update(flag_params)
    return class_params
