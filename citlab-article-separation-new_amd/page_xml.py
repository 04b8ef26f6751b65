"""PAGE-XML reader / writer on ``xml.etree.ElementTree`` (no lxml / cssutils), SURVEY.md row f2.

Re-states the subset of ``python_util/parser/xml/page/{page,page_objects,page_util}.py`` that the two hot-path
pipelines touch: loading text regions / text lines / separator regions with their ``custom`` attributes
(``page.py:300-321,479-590``), setting article ids on text lines (``page_objects.py:426-445``), heading tags
(``heading_net_post_processor.py:175-195``), replacing separator regions (``separator_region_to_page_writer.py:340-358``)
and writing the document back (``page.py:841-851``).  Region objects keep a reference to their DOM node and are
written back in place, so untouched content of the file survives a round trip.

``custom`` attribute grammar: ``name {key:value; key:value;} name2 {...}`` (``page_util.py:5-21``).
"""
import copy
import datetime
import io
import re
import xml.etree.ElementTree as ET

NS_PAGE_XML = "http://schema.primaresearch.org/PAGE/gts/pagecontent/2013-07-15"      # page_constants.py:9
CREATOR = "CITlab"

REGION_TYPES = ("TextRegion", "ImageRegion", "LineDrawingRegion", "GraphicRegion", "TableRegion", "ChartRegion",
                "SeparatorRegion", "MathsRegion", "ChemRegion", "MusicRegion", "AdvertRegion", "NoiseRegion",
                "UnknownRegion")


_SEP_MARK = "AsepSeparatorRegionsHere"


def _escape_attrib(text):
    """xml.etree.ElementTree._escape_attrib"""
    if "&" in text:
        text = text.replace("&", "&amp;")
    if "<" in text:
        text = text.replace("<", "&lt;")
    if ">" in text:
        text = text.replace(">", "&gt;")
    if "\"" in text:
        text = text.replace("\"", "&quot;")
    if "\r" in text:
        text = text.replace("\r", "&#13;")
    if "\n" in text:
        text = text.replace("\n", "&#10;")
    if "\t" in text:
        text = text.replace("\t", "&#09;")
    return text


class PageXmlException(Exception):
    pass


def parse_custom_attr(s):
    """'readingOrder {index:1;} structure {type:article; id:a1;}' -> {'readingOrder': {'index': '1'}, ...}"""
    out = {}
    if not s:
        return out
    for name, body in re.findall(r"([\w-]+)\s*\{([^}]*)\}", s):
        d = out.setdefault(name, {})
        for item in body.split(";"):
            if ":" in item:
                k, v = item.split(":", 1)
                d[k.strip()] = v.strip()
    return out


def format_custom_attr(ddic):
    """page_util.py:5-21."""
    parts = []
    for k1, d2 in ddic.items():
        parts.append("%s {%s}" % (k1, " ".join("%s:%s;" % (k2, v2) for k2, v2 in d2.items())))
    return " ".join(parts)


def parse_points(s):
    """'x1,y1 x2,y2 ...' -> [(x1, y1), ...] (ints)"""
    pts = []
    for tok in (s or "").split():
        x, y = tok.split(",")
        pts.append((int(float(x)), int(float(y))))
    return pts


def format_points(pts):
    return " ".join("%d,%d" % (int(x), int(y)) for x, y in pts)


class Word:
    """page_objects.py:447-550 (subset): id, surrounding polygon, text"""

    def __init__(self, node, page):
        self.node = node
        self.id = node.get("id")
        self.custom = parse_custom_attr(node.get("custom"))
        coords = node.find(page._q("Coords"))
        self.surr_p = parse_points(coords.get("points")) if coords is not None else []
        self.text = page._text_equiv(node)


class TextLine:
    def __init__(self, node, page):
        self.node = node
        self._page = page
        self.id = node.get("id")
        self.custom = parse_custom_attr(node.get("custom"))
        coords = node.find(page._q("Coords"))
        self.surr_p = parse_points(coords.get("points")) if coords is not None else []
        bl = node.find(page._q("Baseline"))
        self.baseline = parse_points(bl.get("points")) if bl is not None else []
        self.text = page._text_equiv(node)
        self.words = [Word(n, page) for n in node.findall(page._q("Word"))]

    def split_copy(self, new_id, points, baseline, words, text):
        """A new TextLine (own DOM node, not yet attached) that carries this line's attributes and custom tags with another
        outline: what ``_create_page_objects`` + ``set_points`` / ``set_baseline`` produce in
        separator_region_to_page_writer.py:133-143.  Children are written in schema order (page_objects.py:317-353):
        Coords, Baseline, Word*, TextEquiv."""
        q = self._page._q
        node = ET.Element(self.node.tag, dict(self.node.attrib))
        node.set("id", new_id)
        ET.SubElement(node, q("Coords"), {"points": format_points(points)})
        if baseline:
            ET.SubElement(node, q("Baseline"), {"points": format_points(baseline)})
        for w in words:
            node.append(copy.deepcopy(w.node))
        if text is not None:
            te = ET.SubElement(node, q("TextEquiv"))
            ET.SubElement(te, q("Unicode")).text = text
        return TextLine(node, self._page)

    def get_article_id(self):
        st = self.custom.get("structure", {})
        return st.get("id") if st.get("type") == "article" else None

    def set_article_id(self, article_id=None):
        """page_objects.py:426-445."""
        if article_id:
            st = self.custom.setdefault("structure", {})
            st["id"] = str(article_id)
            st["type"] = "article"
        else:
            st = self.custom.get("structure")
            if st is not None and "id" in st:
                st.pop("id")
                if not st:
                    self.custom.pop("structure")

    def get_semantic_type(self):
        return self.custom.get("structure", {}).get("semantic_type")

    def set_structure_attribute(self, name, value):
        self.custom.setdefault("structure", {})[name] = str(value)

    def get_bounding_box(self):
        xs = [p[0] for p in self.surr_p]
        ys = [p[1] for p in self.surr_p]
        return min(xs), min(ys), max(xs) - min(xs) + 1, max(ys) - min(ys) + 1     # polygon.py:91-92: width = max-min+1

    def flush(self):
        if self.custom:
            self.node.set("custom", format_custom_attr(self.custom))
        elif "custom" in self.node.attrib:
            del self.node.attrib["custom"]


class Region:
    def __init__(self, node, page, kind):
        self.node = node
        self._page = page
        self.kind = kind
        self.id = node.get("id")
        self.custom = parse_custom_attr(node.get("custom"))
        coords = node.find(page._q("Coords"))
        self.points = parse_points(coords.get("points")) if coords is not None else []

    def get_bounding_box(self):
        xs = [p[0] for p in self.points]
        ys = [p[1] for p in self.points]
        return min(xs), min(ys), max(xs) - min(xs) + 1, max(ys) - min(ys) + 1

    def flush(self):
        if self.custom:
            self.node.set("custom", format_custom_attr(self.custom))


class TextRegion(Region):
    def __init__(self, node, page):
        super().__init__(node, page, "TextRegion")
        self.region_type = node.get("type")
        self.text_lines = [TextLine(n, page) for n in node.findall(page._q("TextLine"))]

    def flush(self):
        super().flush()
        if self.region_type:
            self.node.set("type", self.region_type)
        for tl in self.text_lines:
            tl.flush()

    def replace_text_lines(self, text_lines):
        """The region's TextLine children become exactly ``text_lines`` (kept lines stay where they are, new ones take
        the place of the first removed line or go behind the last line): page.py:682-700 with overwrite=True."""
        q = self._page._q("TextLine")
        old = [n for n in self.node if n.tag == q]
        keep = {id(tl.node) for tl in text_lines}
        pos = None
        for n in old:
            if id(n) not in keep:
                if pos is None:
                    pos = list(self.node).index(n)
                self.node.remove(n)
        if pos is None:
            pos = (list(self.node).index(old[-1]) + 1) if old else len(self.node)
        present = {id(n) for n in self.node}
        for tl in text_lines:
            if id(tl.node) not in present:
                self.node.insert(pos, tl.node)
                pos += 1
            else:
                pos = max(pos, list(self.node).index(tl.node) + 1)
        self.text_lines = list(text_lines)


class SeparatorRegion(Region):
    def __init__(self, node, page):
        super().__init__(node, page, "SeparatorRegion")

    def get_orientation(self):
        return self.custom.get("structure", {}).get("orientation")


class Page:
    """page.py:27-851 (subset)."""

    def __init__(self, path_to_xml=None, creator_name=CREATOR, img_filename=None, img_w=None, img_h=None):
        self.ns = NS_PAGE_XML
        if path_to_xml is not None:
            self.tree = ET.parse(path_to_xml)
            root = self.tree.getroot()
            m = re.match(r"\{(.*)\}", root.tag)
            self.ns = m.group(1) if m else ""
        else:
            self.tree = ET.ElementTree(self._create_document(creator_name, img_filename, img_w or 0, img_h or 0))
        ET.register_namespace("", self.ns)
        self.page_node = self.tree.getroot().find(self._q("Page"))
        if self.page_node is None:
            raise PageXmlException("no <Page> element")

    # The DOM behind two properties: separator regions added by add_separator_region wait as (id, custom, points) records and are spliced into
    # the serialised text by write_page_xml (a page of the separator command line carries thousands of them: ElementTree nodes cost 13 us each to
    # build, indent and serialise, a formatted record 1 us).  Anything that READS the DOM goes through `tree` / `page_node` and finds real nodes.
    @property
    def tree(self):
        self._materialize()
        return self._tree

    @tree.setter
    def tree(self, value):
        self._tree = value

    @property
    def page_node(self):
        self._materialize()
        return self._page_node

    @page_node.setter
    def page_node(self, value):
        self._page_node = value

    def _materialize(self):
        pending = self.__dict__.get("_sep_fast")
        if pending:
            self._sep_fast = []
            for rid, custom, points in pending:
                attrib = {"id": rid}
                if custom:
                    attrib["custom"] = custom
                node = ET.SubElement(self._page_node, self._q("SeparatorRegion"), attrib)
                ET.SubElement(node, self._q("Coords"), {"points": points})

    # -- helpers -------------------------------------------------------------------------------
    def _q(self, tag):
        return "{%s}%s" % (self.ns, tag) if self.ns else tag

    def _text_equiv(self, node):
        te = node.find(self._q("TextEquiv"))
        if te is None:
            return ""
        u = te.find(self._q("Unicode"))
        return (u.text or "") if u is not None else ""

    def _create_document(self, creator, filename, w, h):
        root = ET.Element("{%s}PcGts" % NS_PAGE_XML)
        md = ET.SubElement(root, "{%s}Metadata" % NS_PAGE_XML)
        now = datetime.datetime.now(datetime.timezone.utc).strftime("%Y-%m-%dT%H:%M:%S")
        ET.SubElement(md, "{%s}Creator" % NS_PAGE_XML).text = creator
        ET.SubElement(md, "{%s}Created" % NS_PAGE_XML).text = now
        ET.SubElement(md, "{%s}LastChange" % NS_PAGE_XML).text = now
        ET.SubElement(root, "{%s}Page" % NS_PAGE_XML, {"imageFilename": filename or "", "imageWidth": str(int(w)),
                                                        "imageHeight": str(int(h))})
        return root

    # -- queries -------------------------------------------------------------------------------
    def get_image_resolution(self):
        return int(self._page_node.get("imageWidth")), int(self._page_node.get("imageHeight"))

    def get_text_regions(self, text_region_type=None):
        """Document order of './/TextRegion' (nested regions included), page.py:479-506."""
        # (readers of text regions / lines do not need the waiting separator records as nodes: `_page_node`)
        regs = [TextRegion(n, self) for n in self._page_node.iter(self._q("TextRegion"))]
        if text_region_type:
            regs = [r for r in regs if r.region_type == text_region_type]
        return regs

    def get_regions(self):
        """page.py:528-550: {'TextRegion': [...], 'SeparatorRegion': [...], ...} (only types that occur)."""
        out = {}
        for kind in REGION_TYPES:
            nodes = list(self.page_node.iter(self._q(kind)))
            if not nodes:
                continue
            if kind == "TextRegion":
                out[kind] = [TextRegion(n, self) for n in nodes]
            elif kind == "SeparatorRegion":
                out[kind] = [SeparatorRegion(n, self) for n in nodes]
            else:
                out[kind] = [Region(n, self, kind) for n in nodes]
        return out

    def get_textlines(self):
        return [TextLine(n, self) for n in self._page_node.iter(self._q("TextLine"))]

    def get_ids(self):
        return {n.get("id") for n in self._tree.getroot().iter() if n.get("id")} | {r[0] for r in self.__dict__.get("_sep_fast") or ()}

    def get_unique_id(self, page_object_name, ids=None):
        """page.py:464-477: smallest free '<name>_<n>', n >= 1 (the reference gives up after n = 1000)."""
        if ids is None:
            ids = self.get_ids()
            i = 1
        else:
            # a caller's id set only grows (add_separator_region): the smallest free number never decreases, so the search resumes where
            # the last one ended -- a page with thousands of separator regions paid a quadratic walk here (round 6)
            hints = self.__dict__.setdefault("_id_hints", {})
            i = hints.get((page_object_name, id(ids)), 1)
        while f"{page_object_name}_{i}" in ids:
            i += 1
        if "_id_hints" in self.__dict__ and ids is not None and (page_object_name, id(ids)) in self._id_hints or ids is getattr(self, "_sep_ids", None):
            self.__dict__.setdefault("_id_hints", {})[(page_object_name, id(ids))] = i
        return f"{page_object_name}_{i}"

    # -- modification --------------------------------------------------------------------------
    def set_text_regions(self, text_regions, overwrite=False):
        """Write the (modified) region / line objects back into their DOM nodes (page.py:682-700)."""
        for r in text_regions:
            r.flush()

    def set_textline_attr(self, textlines):
        for tl in textlines:
            tl.flush()

    def remove_regions(self, region_type):
        parents = {c: p for p in self.tree.getroot().iter() for c in p}
        for n in list(self.page_node.iter(self._q(region_type))):
            parents[n].remove(n)
        self._sep_ids = None
        self.__dict__.pop("_id_hints", None)

    def add_separator_region(self, points, orientation):
        """separator_region_to_page_writer.py:340-358: id 'SeparatorRegion_<n>' (the smallest free n), custom structure {orientation:...}.
        The region waits as a record (see `tree`); its id is final."""
        if getattr(self, "_sep_ids", None) is None:
            self._sep_ids = self.get_ids()               # one DOM walk per batch of additions
            self._sep_next = 1
        ids, n = self._sep_ids, self._sep_next
        while f"SeparatorRegion_{n}" in ids:             # (ids only grow: the smallest free number never decreases)
            n += 1
        rid = f"SeparatorRegion_{n}"
        ids.add(rid)
        self._sep_next = n + 1
        cache = self.__dict__.setdefault("_sep_custom", {})
        if orientation not in cache:
            cache[orientation] = format_custom_attr({"structure": {"orientation": orientation}}) if orientation else None
        self.__dict__.setdefault("_sep_fast", []).append((rid, cache[orientation], format_points(points)))
        return rid

    def write_page_xml(self, save_path, creator=CREATOR, comments=None):
        pending = self.__dict__.get("_sep_fast") or []
        tree, page_node = self._tree, self._page_node        # (not the properties: the waiting records are spliced in as text below)
        md = tree.getroot().find(self._q("Metadata"))
        if md is not None:
            lc = md.find(self._q("LastChange"))
            if lc is None:
                lc = ET.SubElement(md, self._q("LastChange"))
            lc.text = datetime.datetime.now(datetime.timezone.utc).strftime("%Y-%m-%dT%H:%M:%S")
        mark = ET.SubElement(page_node, self._q(_SEP_MARK)) if pending else None
        try:
            try:
                ET.indent(tree, space="  ")
            except AttributeError:  # pragma: no cover (python < 3.9)
                pass
            if not pending:
                tree.write(save_path, encoding="UTF-8", xml_declaration=True)
                return
            buf = io.BytesIO()
            tree.write(buf, encoding="UTF-8", xml_declaration=True)
        finally:
            if mark is not None:
                page_node.remove(mark)
        # the records exactly as ElementTree would have written their nodes (children of <Page>: four spaces, their <Coords>: six)
        esc = _escape_attrib
        block = "".join(
            f'    <SeparatorRegion id="{esc(rid)}"' + (f' custom="{esc(custom)}"' if custom else "") +
            f'>\n      <Coords points="{esc(points)}" />\n    </SeparatorRegion>\n' for rid, custom, points in pending).encode("utf-8")
        text = buf.getvalue()
        needle = ("    <%s />\n" % _SEP_MARK).encode("utf-8")
        if text.count(needle) != 1:                          # (a namespace prefix, a missing indent: write it the slow way)
            self._materialize()
            self._tree.write(save_path, encoding="UTF-8", xml_declaration=True)
            return
        with open(save_path, "wb") as f:
            f.write(text.replace(needle, block, 1))
