"""Writers that put region information coming out of the ARU-Net post-processors into PAGE-XML.

    RegionToPageWriter             region_to_page_writer.py:13-46
    SeparatorRegionToPageWriter    separator_region_to_page_writer.py:11-25,107-386

Deviation (documented in DESIGN.md): ``merge_regions`` adds the separator regions but does not split the text lines
a vertical separator runs through -- the reference does that with shapely's polygon difference, and shapely/GEOS is
not available to this build.  Polygons with holes are written by their exterior ring (the reference additionally
cuts a polygon at holes larger than 1000 px^2, ``:329-337``).
"""
import os

from .image_io import get_image_dimensions
from .net_post_processing_helper import get_scaling_factor
from .page_xml import Page

SEPARATOR_REGION = "SeparatorRegion"


class RegionToPageWriter:
    """Owns the Page object a post-processor writes its regions into (``region_to_page_writer.py:13-46``)."""

    def __init__(self, path_to_page, path_to_image=None, fixed_height=None, scaling_factor=None, *args, **kwargs):
        self.path_to_page = path_to_page
        self.scaling_factor = None
        size = None
        if path_to_image is not None:
            size = get_image_dimensions(path_to_image)          # (width, height)
            self.scaling_factor = get_scaling_factor(size[1], size[0], scaling_factor, fixed_height)
        self.page_object = self.load_page_object(path_to_page, path_to_image, size)

    def load_page_object(self, path_to_page, path_to_image, size=None):
        """An existing PAGE file is parsed; a missing one becomes an empty page that names the image and carries the
        image size multiplied by the scaling factor and truncated (the reference writes the
        *scaled* size there, kept)."""
        if os.path.exists(path_to_page):
            return Page(path_to_page)
        width, height = size if size is not None else get_image_dimensions(path_to_image)
        return Page(img_filename=path_to_image, img_w=int(self.scaling_factor * width),
                    img_h=int(self.scaling_factor * height))

    def save_page_xml(self, save_path):
        folder = os.path.dirname(save_path)
        if folder and not os.path.isdir(folder):
            os.makedirs(folder, exist_ok=True)
        self.page_object.write_page_xml(save_path)


class SeparatorRegionToPageWriter(RegionToPageWriter):
    def __init__(self, path_to_page, path_to_image=None, fixed_height=None, scaling_factor=None, region_dict=None):
        super().__init__(path_to_page, path_to_image, fixed_height, scaling_factor)
        self.region_dict = region_dict or {}

    def remove_separator_regions_from_page(self):
        self.page_object.remove_regions(SEPARATOR_REGION)

    def merge_regions(self, remove_holes=True):
        """:359-386 order: plain, horizontal, vertical; every polygon becomes one SeparatorRegion whose custom tag
        carries the orientation."""
        for separator_type in (SEPARATOR_REGION, SEPARATOR_REGION + "_horizontal", SEPARATOR_REGION + "_vertical"):
            polygons = self.region_dict.get(separator_type)
            if polygons is None:
                continue
            orientation = separator_type[len(SEPARATOR_REGION) + 1:] or None
            for polygon in polygons:
                rings = polygon if (polygon and isinstance(polygon[0], (list, tuple))
                                    and polygon[0] and isinstance(polygon[0][0], (list, tuple))) else [polygon]
                self.page_object.add_separator_region(rings[0], orientation)
