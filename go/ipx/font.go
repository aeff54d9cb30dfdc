package ipx

/*
#include <stdlib.h>
#include "ipx.h"
*/
import "C"

import (
	"image"
	"unsafe"
)

// Font is truetype.Parse + freetype.Context.DrawString inside the library (ipx_font_*): what NewWatermarker holds
// (operations/watermark.go:25-38), without hinting -- the reference never enables it.
type Font struct{ c *C.ipx_font }

// ParseFont is truetype.Parse(goregular.TTF) (watermark.go:30-31).  An error is the reference's nil font: "font not loaded".
func ParseFont(ttf []byte) (*Font, error) {
	var f *C.ipx_font
	if err := call(func() C.int { return C.ipx_font_create(ptr(ttf), C.size_t(len(ttf)), &f) }); err != nil {
		return nil, err
	}
	return &Font{f}, nil
}

func (f *Font) Close() { C.ipx_font_destroy(f.c); f.c = nil }

// TextWidthPx is int(textWidth.Ceil()) of watermark.go:108-117 (sum of advances, no kerning).
func (f *Font) TextWidthPx(text string, size float64) (int, error) {
	ct := C.CString(text)
	defer C.free(unsafe.Pointer(ct))
	var px C.int
	err := call(func() C.int { return C.ipx_font_text_width(f.c, ct, C.double(size), nil, &px) })
	return int(px), err
}

// DrawString returns the DrawMask calls of c.DrawString(text, freetype.Pt(px, py)) with c.SetClip(image.Rect(0, 0, w, h))
// (watermark.go:98-104,151) as Glyphs whose masks are Go copies.
func (f *Font) DrawString(text string, size float64, px, py, w, h int) ([]Glyph, error) {
	ct := C.CString(text)
	defer C.free(unsafe.Pointer(ct))
	var gl *C.ipx_glyph
	var n C.int
	if err := call(func() C.int {
		rc := C.ipx_font_draw_string(f.c, ct, C.double(size), C.int(px), C.int(py), C.int(w), C.int(h), &gl, &n, nil)
		return rc
	}); err != nil {
		return nil, err
	}
	out := make([]Glyph, int(n))
	cg := unsafe.Slice(gl, int(n))
	for i, g := range cg {
		m := image.NewAlpha(image.Rect(0, 0, int(g.mw), int(g.mh)))
		for y := 0; y < int(g.mh); y++ {
			row := unsafe.Slice((*byte)(unsafe.Add(unsafe.Pointer(g.mask), y*int(g.mstride))), int(g.mw))
			copy(m.Pix[y*m.Stride:], row)
		}
		out[i] = Glyph{Mask: m, Dr: image.Rect(int(g.dr.x0), int(g.dr.y0), int(g.dr.x1), int(g.dr.y1)), Mp: image.Pt(int(g.mpx), int(g.mpy))}
	}
	// NOTE: the glyph list is valid until the next call on this OS thread: call() keeps the thread locked only for the duration of the
	// C call, so copy inside one locked section when goroutines share fonts heavily (runtime.LockOSThread around DrawString).
	return out, nil
}
