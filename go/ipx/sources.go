package ipx

/*
#include "ipx.h"
*/
import "C"

import (
	"fmt"
	"image"
	"image/color"
)

// The entries for the other image types image.Decode returns (image_processor.go:47: PNG and GIF uploads, grey and four-component
// JPEGs).  Each takes n frames of ONE type and size, their Pix rows packed one frame after the other (Pack* below), and gives the
// bytes the reference's helpers produce on an image of that type: resizeImage interpolates the type's own 16-bit taps, the crop
// thumbnail and the watermark go through image/draw's conversion to RGBA8 first.

// RunHostNRGBA: *image.NRGBA frames (PNGs with alpha), 4 bytes per pixel, not premultiplied.
func (p *Plan) RunHostNRGBA(n int, pix, resizeOut, thumbOut, wmOut []byte) error {
	i := p.Info
	return call(func() C.int {
		return C.ipx_plan_run_host_nrgba(p.x.c, p.c, C.int(n), ptr(pix), C.int(p.w*4), C.size_t(p.w*p.h*4),
			ptr(resizeOut), C.size_t(i.ResizeBytes), ptr(thumbOut), C.size_t(i.ThumbBytes), ptr(wmOut), C.size_t(i.WmBytes))
	})
}

// RunHostGray: *image.Gray frames (one-component JPEGs, 8-bit grey PNGs), 1 byte per pixel.
func (p *Plan) RunHostGray(n int, pix, resizeOut, thumbOut, wmOut []byte) error {
	i := p.Info
	return call(func() C.int {
		return C.ipx_plan_run_host_gray(p.x.c, p.c, C.int(n), ptr(pix), C.int(p.w), C.size_t(p.w*p.h),
			ptr(resizeOut), C.size_t(i.ResizeBytes), ptr(thumbOut), C.size_t(i.ThumbBytes), ptr(wmOut), C.size_t(i.WmBytes))
	})
}

// RunHostPaletted: *image.Paletted frames (GIF uploads, palette PNGs): one index byte per pixel and, per frame, 256 entries of
// (R, G, B, A) as PackPalette writes them.
func (p *Plan) RunHostPaletted(n int, index, palettes, resizeOut, thumbOut, wmOut []byte) error {
	i := p.Info
	return call(func() C.int {
		return C.ipx_plan_run_host_paletted(p.x.c, p.c, C.int(n), ptr(index), C.int(p.w), C.size_t(p.w*p.h), ptr(palettes),
			ptr(resizeOut), C.size_t(i.ResizeBytes), ptr(thumbOut), C.size_t(i.ThumbBytes), ptr(wmOut), C.size_t(i.WmBytes))
	})
}

// Deep is one of the image types that reach x/image's generic scaleX_Image and image/draw's drawRGBA / drawCMYK.
type Deep int

const (
	NRGBA64 Deep = C.IPX_DEEP_NRGBA64 // 16-bit truecolour or grey PNG with alpha / tRNS
	RGBA64  Deep = C.IPX_DEEP_RGBA64  // 16-bit truecolour PNG
	Gray16  Deep = C.IPX_DEEP_GRAY16  // 16-bit grey PNG
	CMYK    Deep = C.IPX_DEEP_CMYK    // four-component JPEG
)

func (d Deep) bytesPerPixel() int {
	switch d {
	case Gray16:
		return 2
	case CMYK:
		return 4
	}
	return 8
}

// RunHostDeep: frames of a Deep type, Pix as Go holds it (big-endian 16-bit channels; C M Y K bytes).
func (p *Plan) RunHostDeep(n int, kind Deep, pix, resizeOut, thumbOut, wmOut []byte) error {
	i, bpp := p.Info, kind.bytesPerPixel()
	return call(func() C.int {
		return C.ipx_plan_run_host_deep(p.x.c, p.c, C.int(n), C.int(kind), ptr(pix), C.int(p.w*bpp), C.size_t(p.w*p.h*bpp),
			ptr(resizeOut), C.size_t(i.ResizeBytes), ptr(thumbOut), C.size_t(i.ThumbBytes), ptr(wmOut), C.size_t(i.WmBytes))
	})
}

// packRows appends the rows of one frame without their stride padding (a sub-image or a decoder's padded rows).
func packRows(dst, pix []byte, stride, rowBytes, h int) []byte {
	for y := 0; y < h; y++ {
		dst = append(dst, pix[y*stride:y*stride+rowBytes]...)
	}
	return dst
}

// PackPalette appends the 1024 bytes RunHostPaletted wants for one frame: Palette[i] as non-premultiplied (R, G, B, A), missing
// entries zero.  The entries the GIF and PNG decoders produce (opaque color.RGBA, the zero colour of a GIF's transparent index,
// color.NRGBA for a PNG's tRNS) convert without loss; ok = false for a palette with any other colour type (keep the CPU path).
func PackPalette(dst []byte, pal color.Palette) ([]byte, bool) {
	var e [1024]byte
	if len(pal) > 256 {
		return dst, false
	}
	for i, c := range pal {
		switch c := c.(type) {
		case color.RGBA:
			if c.A != 0xff && c != (color.RGBA{}) {
				return dst, false // a premultiplied translucent entry: no exact NRGBA8 form in general
			}
			e[4*i], e[4*i+1], e[4*i+2], e[4*i+3] = c.R, c.G, c.B, c.A
		case color.NRGBA:
			e[4*i], e[4*i+1], e[4*i+2], e[4*i+3] = c.R, c.G, c.B, c.A
		default:
			return dst, false
		}
	}
	return append(dst, e[:]...), true
}

// RunImages dispatches a batch of decoded images of one concrete type and size (the plan's) to the entry for that type -- the type
// switch the worker needs after image.Decode.  ErrUnsupported-style errors (IsUnsupported) mean: process these with the CPU path.
func (p *Plan) RunImages(imgs []image.Image, resizeOut, thumbOut, wmOut []byte) error {
	if len(imgs) == 0 {
		return nil
	}
	n := len(imgs)
	for _, im := range imgs {
		if b := im.Bounds(); b.Dx() != p.w || b.Dy() != p.h || b.Min != (image.Point{}) {
			return &Error{Unsupported, "frame bounds differ from the plan's"}
		}
	}
	mixed := &Error{Unsupported, "a batch holds one image type"}
	var pix []byte
	switch first := imgs[0].(type) {
	case *image.RGBA:
		for _, im := range imgs {
			m, ok := im.(*image.RGBA)
			if !ok {
				return mixed
			}
			pix = packRows(pix, m.Pix, m.Stride, p.w*4, p.h)
		}
		return p.RunHost(n, pix, resizeOut, thumbOut, wmOut)
	case *image.NRGBA:
		for _, im := range imgs {
			m, ok := im.(*image.NRGBA)
			if !ok {
				return mixed
			}
			pix = packRows(pix, m.Pix, m.Stride, p.w*4, p.h)
		}
		return p.RunHostNRGBA(n, pix, resizeOut, thumbOut, wmOut)
	case *image.Gray:
		for _, im := range imgs {
			m, ok := im.(*image.Gray)
			if !ok {
				return mixed
			}
			pix = packRows(pix, m.Pix, m.Stride, p.w, p.h)
		}
		return p.RunHostGray(n, pix, resizeOut, thumbOut, wmOut)
	case *image.Paletted:
		var pals []byte
		for _, im := range imgs {
			m, ok := im.(*image.Paletted)
			if !ok {
				return mixed
			}
			pix = packRows(pix, m.Pix, m.Stride, p.w, p.h)
			if pals, ok = PackPalette(pals, m.Palette); !ok {
				return &Error{Unsupported, "palette entries of a colour type the GPU path does not expand"}
			}
		}
		return p.RunHostPaletted(n, pix, pals, resizeOut, thumbOut, wmOut)
	case *image.YCbCr:
		var y, cb, cr []byte
		cw, ch := first.CStride, len(first.Cb)/first.CStride
		for _, im := range imgs {
			m, ok := im.(*image.YCbCr)
			if !ok || m.SubsampleRatio != first.SubsampleRatio || m.YStride != first.YStride || m.CStride != first.CStride {
				return mixed
			}
			y = append(y, m.Y[:m.YStride*p.h]...)
			cb = append(cb, m.Cb[:cw*ch]...)
			cr = append(cr, m.Cr[:cw*ch]...)
		}
		return p.RunHostYCbCr(n, first, y, cb, cr, resizeOut, thumbOut, wmOut)
	case *image.NRGBA64:
		for _, im := range imgs {
			m, ok := im.(*image.NRGBA64)
			if !ok {
				return mixed
			}
			pix = packRows(pix, m.Pix, m.Stride, p.w*8, p.h)
		}
		return p.RunHostDeep(n, NRGBA64, pix, resizeOut, thumbOut, wmOut)
	case *image.RGBA64:
		for _, im := range imgs {
			m, ok := im.(*image.RGBA64)
			if !ok {
				return mixed
			}
			pix = packRows(pix, m.Pix, m.Stride, p.w*8, p.h)
		}
		return p.RunHostDeep(n, RGBA64, pix, resizeOut, thumbOut, wmOut)
	case *image.Gray16:
		for _, im := range imgs {
			m, ok := im.(*image.Gray16)
			if !ok {
				return mixed
			}
			pix = packRows(pix, m.Pix, m.Stride, p.w*2, p.h)
		}
		return p.RunHostDeep(n, Gray16, pix, resizeOut, thumbOut, wmOut)
	case *image.CMYK:
		for _, im := range imgs {
			m, ok := im.(*image.CMYK)
			if !ok {
				return mixed
			}
			pix = packRows(pix, m.Pix, m.Stride, p.w*4, p.h)
		}
		return p.RunHostDeep(n, CMYK, pix, resizeOut, thumbOut, wmOut)
	}
	return &Error{Unsupported, fmt.Sprintf("no GPU entry for %T", imgs[0])}
}
