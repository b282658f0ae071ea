"""image_transformation_amd: MI355X-native deterministic compositor for the macro-placement
pipeline of FelixMul/image_transformation (its pixel hot path only).

Drop-in call surface (same names, arguments and error behaviour as the reference):

    from image_transformation_amd.compositor import composite, load_object_images, render
    from image_transformation_amd.background_resizing import fill_solid
    from image_transformation_amd.contact_sheet import build_labeled_contact_sheet
    from image_transformation_amd.layout_constraints import compute_canvas_size, parse_ratio

Importing this package does not touch the GPU; the first call into the compositor loads
libmic.so and creates a context on the current ROCm device, or raises -- there is no CPU path.
"""
__version__ = "0.1.0"
