"""Mirror of the reference's ``superresolution_scripts`` package (augmentation_utils, superresolution,
optimizer, superres_utils) on hand-written gfx950 kernels."""
