"""TEST INFRASTRUCTURE: an encoder for TensorFlow ``GraphDef`` files that is independent of the product's reader.

``google.protobuf`` (installed in the image) serialises messages built from descriptors declared HERE with the field
numbers of tensorflow/core/framework/{graph,node_def,attr_value,tensor,tensor_shape,versions}.proto -- including
fields the product's reader does not decode (``device``, ``versions``, ``library``, ``half_val``, ``unknown_rank``,
``experimental_debug_info``), so that "unknown fields are skipped" is exercised by a real encoder.  TensorFlow itself is
not needed: a frozen graph is nothing but this message.

    build_messages() -> namespace with GraphDef / NodeDef / AttrValue / TensorProto / TensorShapeProto classes
    node(...)         -> NodeDef from python values (attrs: int / float / bool / bytes / list / ndarray / shape / dtype)
"""
import numpy as np
from google.protobuf import descriptor_pb2, descriptor_pool, message_factory

F = descriptor_pb2.FieldDescriptorProto
DT_FLOAT, DT_DOUBLE, DT_INT32, DT_STRING, DT_INT64, DT_BOOL, DT_HALF = 1, 2, 3, 7, 9, 10, 19
_NP2DT = {np.dtype("float32"): DT_FLOAT, np.dtype("float64"): DT_DOUBLE, np.dtype("int32"): DT_INT32,
          np.dtype("int64"): DT_INT64}


def _field(msg, name, number, ftype, label=F.LABEL_OPTIONAL, type_name=None, packed=None):
    f = msg.field.add()
    f.name, f.number, f.type, f.label = name, number, ftype, label
    if type_name:
        f.type_name = type_name
    if packed is not None:
        f.options.packed = packed
    return f


_CACHE = {}


def build_messages(packed_repeated=True):
    """``packed_repeated=False`` emits repeated scalars one tag per element (proto2 style, legal on the wire)."""
    if packed_repeated in _CACHE:
        return _CACHE[packed_repeated]
    pkg = "asep_test_tf_%s" % ("p" if packed_repeated else "u")
    fd = descriptor_pb2.FileDescriptorProto()
    fd.name, fd.package, fd.syntax = pkg + ".proto", pkg, "proto3"
    R = F.LABEL_REPEATED

    dim = descriptor_pb2.DescriptorProto(name="Dim")
    _field(dim, "size", 1, F.TYPE_INT64)
    _field(dim, "name", 2, F.TYPE_STRING)
    shape = fd.message_type.add()
    shape.name = "TensorShapeProto"
    shape.nested_type.add().CopyFrom(dim)
    _field(shape, "dim", 2, F.TYPE_MESSAGE, R, f".{pkg}.TensorShapeProto.Dim")
    _field(shape, "unknown_rank", 3, F.TYPE_BOOL)

    tensor = fd.message_type.add()
    tensor.name = "TensorProto"
    _field(tensor, "dtype", 1, F.TYPE_INT32)
    _field(tensor, "tensor_shape", 2, F.TYPE_MESSAGE, type_name=f".{pkg}.TensorShapeProto")
    _field(tensor, "version_number", 3, F.TYPE_INT32)
    _field(tensor, "tensor_content", 4, F.TYPE_BYTES)
    _field(tensor, "float_val", 5, F.TYPE_FLOAT, R, packed=packed_repeated)
    _field(tensor, "double_val", 6, F.TYPE_DOUBLE, R, packed=packed_repeated)
    _field(tensor, "int_val", 7, F.TYPE_INT32, R, packed=packed_repeated)
    _field(tensor, "string_val", 8, F.TYPE_BYTES, R)
    _field(tensor, "int64_val", 10, F.TYPE_INT64, R, packed=packed_repeated)
    _field(tensor, "bool_val", 11, F.TYPE_BOOL, R, packed=packed_repeated)
    _field(tensor, "half_val", 13, F.TYPE_INT32, R, packed=packed_repeated)

    attr = fd.message_type.add()
    attr.name = "AttrValue"
    lst = attr.nested_type.add()
    lst.name = "ListValue"
    _field(lst, "s", 2, F.TYPE_BYTES, R)
    _field(lst, "i", 3, F.TYPE_INT64, R, packed=packed_repeated)
    _field(lst, "f", 4, F.TYPE_FLOAT, R, packed=packed_repeated)
    _field(lst, "b", 5, F.TYPE_BOOL, R, packed=packed_repeated)
    _field(lst, "type", 6, F.TYPE_INT32, R, packed=packed_repeated)
    _field(lst, "shape", 7, F.TYPE_MESSAGE, R, f".{pkg}.TensorShapeProto")
    _field(lst, "tensor", 8, F.TYPE_MESSAGE, R, f".{pkg}.TensorProto")
    _field(attr, "list", 1, F.TYPE_MESSAGE, type_name=f".{pkg}.AttrValue.ListValue")
    _field(attr, "s", 2, F.TYPE_BYTES)
    _field(attr, "i", 3, F.TYPE_INT64)
    _field(attr, "f", 4, F.TYPE_FLOAT)
    _field(attr, "b", 5, F.TYPE_BOOL)
    _field(attr, "type", 6, F.TYPE_INT32)
    _field(attr, "shape", 7, F.TYPE_MESSAGE, type_name=f".{pkg}.TensorShapeProto")
    _field(attr, "tensor", 8, F.TYPE_MESSAGE, type_name=f".{pkg}.TensorProto")
    _field(attr, "placeholder", 9, F.TYPE_STRING)
    for f in attr.field:                       # the real message is a oneof: presence must be explicit (i = 0, b = false)
        if f.name != "list" and f.type != F.TYPE_MESSAGE:
            f.proto3_optional = True
    for k, f in enumerate([f for f in attr.field if f.proto3_optional]):
        attr.oneof_decl.add().name = "_" + f.name
        f.oneof_index = k

    nd = fd.message_type.add()
    nd.name = "NodeDef"
    entry = nd.nested_type.add()
    entry.name = "AttrEntry"
    entry.options.map_entry = True
    _field(entry, "key", 1, F.TYPE_STRING)
    _field(entry, "value", 2, F.TYPE_MESSAGE, type_name=f".{pkg}.AttrValue")
    dbg = nd.nested_type.add()
    dbg.name = "ExperimentalDebugInfo"
    _field(dbg, "original_node_names", 1, F.TYPE_STRING, R)
    _field(nd, "name", 1, F.TYPE_STRING)
    _field(nd, "op", 2, F.TYPE_STRING)
    _field(nd, "input", 3, F.TYPE_STRING, R)
    _field(nd, "device", 4, F.TYPE_STRING)
    _field(nd, "attr", 5, F.TYPE_MESSAGE, R, f".{pkg}.NodeDef.AttrEntry")
    _field(nd, "experimental_debug_info", 6, F.TYPE_MESSAGE, type_name=f".{pkg}.NodeDef.ExperimentalDebugInfo")

    ver = fd.message_type.add()
    ver.name = "VersionDef"
    _field(ver, "producer", 1, F.TYPE_INT32)
    _field(ver, "min_consumer", 2, F.TYPE_INT32)
    _field(ver, "bad_consumers", 3, F.TYPE_INT32, R, packed=packed_repeated)

    lib = fd.message_type.add()
    lib.name = "FunctionDefLibrary"
    _field(lib, "opaque", 1, F.TYPE_BYTES, R)

    gd = fd.message_type.add()
    gd.name = "GraphDef"
    _field(gd, "node", 1, F.TYPE_MESSAGE, R, f".{pkg}.NodeDef")
    _field(gd, "library", 2, F.TYPE_MESSAGE, type_name=f".{pkg}.FunctionDefLibrary")
    _field(gd, "version", 3, F.TYPE_INT32)
    _field(gd, "versions", 4, F.TYPE_MESSAGE, type_name=f".{pkg}.VersionDef")

    pool = descriptor_pool.DescriptorPool()
    pool.Add(fd)

    class NS:
        pass
    ns = NS()
    for name in ("GraphDef", "NodeDef", "AttrValue", "TensorProto", "TensorShapeProto"):
        setattr(ns, name, message_factory.GetMessageClass(pool.FindMessageTypeByName(f"{pkg}.{name}")))
    _CACHE[packed_repeated] = ns
    return ns


class Shape:
    """attr value: a TensorShapeProto (dims may be -1 = unknown)"""
    def __init__(self, *dims):
        self.dims = dims


class DType:
    def __init__(self, enum):
        self.enum = enum


def fill_tensor(ns, t, arr, encoding="content"):
    """encoding: 'content' (tensor_content bytes), 'repeated' (float_val / int_val ...), 'splat' (one repeated value
    for a constant tensor)"""
    arr = np.asarray(arr)
    t.dtype = _NP2DT[arr.dtype]
    for d in arr.shape:
        t.tensor_shape.dim.add().size = int(d)
    if encoding == "content":
        t.tensor_content = np.ascontiguousarray(arr).tobytes()
        return
    field = {DT_FLOAT: t.float_val, DT_DOUBLE: t.double_val, DT_INT32: t.int_val, DT_INT64: t.int64_val}[t.dtype]
    flat = arr.reshape(-1).tolist()
    if encoding == "splat":
        assert len(set(flat)) <= 1
        flat = flat[:1]
    field.extend(flat)


def node(ns, name, op, inputs=(), device="", debug=False, tensor_encoding="content", **attrs):
    n = ns.NodeDef(name=name, op=op, input=list(inputs), device=device)
    if debug:
        n.experimental_debug_info.original_node_names.append(name + "_orig")
    for key, val in attrs.items():
        a = n.attr[key]
        if isinstance(val, bool):
            a.b = val
        elif isinstance(val, int):
            a.i = val
        elif isinstance(val, float):
            a.f = val
        elif isinstance(val, (bytes, str)):
            a.s = val.encode() if isinstance(val, str) else val
        elif isinstance(val, DType):
            a.type = val.enum
        elif isinstance(val, Shape):
            for d in val.dims:
                a.shape.dim.add().size = d
        elif isinstance(val, np.ndarray):
            fill_tensor(ns, a.tensor, val, tensor_encoding)
        elif isinstance(val, (list, tuple)):
            if all(isinstance(v, bool) for v in val) and val:
                a.list.b.extend(val)
            elif all(isinstance(v, int) for v in val):
                a.list.i.extend(val)
                if not val:
                    a.list.SetInParent()
            elif all(isinstance(v, float) for v in val):
                a.list.f.extend(val)
            else:
                a.list.s.extend(v.encode() if isinstance(v, str) else v for v in val)
        else:
            raise TypeError(f"attr {key}: {type(val)}")
    return n


def graphdef(ns, nodes, producer=134):
    g = ns.GraphDef()
    g.node.extend(nodes)
    g.versions.producer = producer
    g.versions.min_consumer = 12
    g.library.SetInParent()
    return g
