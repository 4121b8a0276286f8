"""Image file reading under the reference's name (auromat/util/image.py:17-39, which goes through skimage.io)."""
import numpy as np


def loadImage(imagePath):
    """
    Return the RGB image in its native range ([0, 255] for uint8, [0, 65535] for uint16), shape (height, width, 3);
    grey images are repeated over the three channels, an alpha channel is ignored.  ``.npy`` arrays are returned as
    stored; everything else is read with Pillow.  Not meant for RAW files (the reference says the same: develop them
    with rawpy first).
    """
    if imagePath.lower().endswith('.npy'):
        rgb = np.load(imagePath)
    else:
        try:
            from PIL import Image
        except ImportError:
            raise NotImplementedError('Reading ' + imagePath + ' needs Pillow; pass the image as an array instead')
        with Image.open(imagePath) as im:
            if im.mode in ('I;16', 'I;16B', 'I;16L', 'I'):
                rgb = np.asarray(im).astype(np.uint16)
            else:
                if im.mode not in ('RGB', 'L'):
                    im = im.convert('RGB')
                rgb = np.asarray(im)
    if rgb.ndim == 2:
        rgb = np.repeat(rgb[:, :, None], 3, axis=2)
    rgb = np.ascontiguousarray(rgb[:, :, :3])
    assert rgb.ndim == 3 and rgb.shape[2] == 3, imagePath + '; wrong shape: ' + str(rgb.shape)
    return rgb


__all__ = ['loadImage']
