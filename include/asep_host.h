/* Host-side helper of the image decode in front of the GPU path (citlab-article-separation-new_amd/csrc/host_png.c -> libasep_host.so,
 * plain C, no GPU): what cv2.imread does with libpng for the scans the pipelines read
 * (net_post_processing_helper.py:29, swt_dist_trafo.py:19, input_dataset.py:279-280).  image_io.py binds it with ctypes and
 * falls back to Pillow for every PNG flavour that is not handled here. */
#ifndef ASEP_HOST_H
#define ASEP_HOST_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* Reverse the PNG row filters (PNG specification 9.2: None, Sub, Up, Average, Paeth) of a non-interlaced image with 8-bit samples.
 * filtered: rows x (1 + stride) bytes as they come out of the zlib stream (filter type byte + filtered row); out: rows x stride
 * bytes; bpp = bytes per pixel (1 gray, 2 gray + alpha, 3 RGB, 4 RGBA).  Returns 0, or -(row + 1) for an unknown filter type. */
long asep_png_unfilter(const uint8_t* filtered, long rows, long stride, int bpp, uint8_t* out);

/* out[i] = {in[3i+2], in[3i+1], in[3i]}: RGB rows -> BGR (cv2's channel order), n pixels. */
void asep_rgb_to_bgr(const uint8_t* in, size_t n, uint8_t* out);

#ifdef __cplusplus
}
#endif
#endif
