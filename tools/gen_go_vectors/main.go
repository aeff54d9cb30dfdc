// gen_go_vectors turns "parity unpinned" into one command for a maintainer who has a Go toolchain.
//
// The reference's pixel arithmetic lives in modules that are not vendored (golang.org/x/image v0.33.0, Go's image/draw and image/jpeg,
// github.com/golang/freetype e2365dfdc4a0) and the build image of this repository has no Go, so the committed known answers
// (tests/golden/kats.json) were derived by hand and by a Python model.  This program feeds the SAME inputs to the real libraries
// and writes tests/golden/kats_go.json in the same format (origin "go"); the test suite loads every tests/golden/kats*.json, so
// the oracle and the GPU path are then checked against Go's own output without any code change.  It also prints every case whose
// Go answer differs from the committed expectation.
//
//	cd tools/gen_go_vectors && go run . -in ../../tests/golden/kats.json -out ../../tests/golden/kats_go.json \
//	    [-font /path/to/Go-Regular.ttf -fontout ../../tests/golden/font_go.json] [-jpegout ../../tests/golden/jpeg_go.json]
//
// SOURCE ONLY: it has never been compiled (no Go toolchain in the build image).
package main

import (
	"bytes"
	"encoding/hex"
	"encoding/json"
	"flag"
	"fmt"
	"image"
	"image/color"
	"image/draw"
	"image/jpeg"
	"math/rand"
	"os"
	"reflect"

	"github.com/golang/freetype"
	"github.com/golang/freetype/truetype"
	xdraw "golang.org/x/image/draw"
	"golang.org/x/image/font/gofont/goregular"
)

type kats struct {
	Note  string                   `json:"note"`
	Cases []map[string]interface{} `json:"cases"`
}

func ints(v interface{}) []int {
	a := v.([]interface{})
	out := make([]int, len(a))
	for i, x := range a {
		out[i] = int(x.(float64))
	}
	return out
}

func byteSlice(v interface{}) []uint8 {
	a := v.([]interface{})
	out := make([]uint8, len(a))
	for i, x := range a {
		out[i] = uint8(x.(float64))
	}
	return out
}

func toList(b []uint8) []interface{} {
	out := make([]interface{}, len(b))
	for i, x := range b {
		out[i] = float64(x)
	}
	return out
}

func rect(v interface{}) image.Rectangle {
	r := ints(v)
	return image.Rect(r[0], r[1], r[2], r[3])
}

func rgba(pix interface{}, w, h int) *image.RGBA {
	m := image.NewRGBA(image.Rect(0, 0, w, h))
	copy(m.Pix, byteSlice(pix))
	return m
}

func nrgba(pix interface{}, w, h int) *image.NRGBA {
	m := image.NewNRGBA(image.Rect(0, 0, w, h))
	copy(m.Pix, byteSlice(pix))
	return m
}

// image.YCbCrSubsampleRatio numbering of include/ipx.h: 0 = 4:4:4, 1 = 4:2:2, 2 = 4:2:0, 3 = 4:4:0
func ycbcr(img map[string]interface{}) *image.YCbCr {
	w, h := int(img["w"].(float64)), int(img["h"].(float64))
	ratios := []image.YCbCrSubsampleRatio{image.YCbCrSubsampleRatio444, image.YCbCrSubsampleRatio422, image.YCbCrSubsampleRatio420, image.YCbCrSubsampleRatio440}
	m := image.NewYCbCr(image.Rect(0, 0, w, h), ratios[int(img["ratio"].(float64))])
	copy(m.Y, byteSlice(img["y"]))
	copy(m.Cb, byteSlice(img["cb"]))
	copy(m.Cr, byteSlice(img["cr"]))
	return m
}

func op(v interface{}) draw.Op { // include/ipx.h: IPX_OP_OVER = 0, IPX_OP_SRC = 1
	if int(v.(float64)) == 1 {
		return draw.Src
	}
	return draw.Over
}

func num(c map[string]interface{}, k string) int { return int(c[k].(float64)) }

// the Go answer for one case, or nil for kinds that restate the reference's own (in-repository) rules
func answer(c map[string]interface{}) interface{} {
	switch c["kind"].(string) {
	case "scale", "scale_nrgba", "scale_ycbcr":
		dst := rgba(c["dst"], num(c, "dw"), num(c, "dh"))
		var src image.Image
		switch c["kind"].(string) {
		case "scale":
			src = rgba(c["src"], num(c, "sw"), num(c, "sh"))
		case "scale_nrgba":
			src = nrgba(c["src"], num(c, "sw"), num(c, "sh"))
		default:
			src = ycbcr(c["img"].(map[string]interface{}))
		}
		o := xdraw.Over // a YCbCr case has no op: resizeImage always passes Over (resize.go:123)
		if v, ok := c["op"]; ok && int(v.(float64)) == 1 {
			o = xdraw.Src
		}
		xdraw.BiLinear.Scale(dst, rect(c["dr"]), src, rect(c["sr"]), o, nil) // resize.go:123, thumbnail.go:129
		return toList(dst.Pix)
	case "draw", "draw_nrgba", "draw_ycbcr":
		dst := rgba(c["dst"], num(c, "dw"), num(c, "dh"))
		var src image.Image
		o := draw.Src
		switch c["kind"].(string) {
		case "draw":
			src, o = rgba(c["src"], num(c, "sw"), num(c, "sh")), op(c["op"])
		case "draw_nrgba":
			src, o = nrgba(c["src"], num(c, "sw"), num(c, "sh")), op(c["op"])
		default:
			src = ycbcr(c["img"].(map[string]interface{}))
		}
		sp := ints(c["sp"])
		draw.Draw(dst, rect(c["r"]), src, image.Pt(sp[0], sp[1]), o) // watermark.go:92
		return toList(dst.Pix)
	case "glyphs":
		dst := rgba(c["dst"], num(c, "dw"), num(c, "dh"))
		col := byteSlice(c["col"])
		uni := image.NewUniform(color.RGBA{col[0], col[1], col[2], col[3]}) // parseColor's color.RGBA, NOT premultiplied (watermark.go:183-185)
		for _, gv := range c["glyphs"].([]interface{}) {
			g := gv.(map[string]interface{})
			mw, mh := int(g["mw"].(float64)), int(g["mh"].(float64))
			mask := image.NewAlpha(image.Rect(0, 0, mw, mh))
			copy(mask.Pix, byteSlice(g["mask"]))
			mp := ints(g["mp"])
			draw.DrawMask(dst, rect(g["dr"]), uni, image.Point{}, mask, image.Pt(mp[0], mp[1]), draw.Over) // what freetype's DrawString calls
		}
		return toList(dst.Pix)
	}
	return nil
}

// ---- the glyph mask producer: freetype.Context.DrawString onto a constant frame.  The frame after the call is what
// ipx_font_draw_string + the glyph composite must reproduce (tests/test_go_vectors.py); the face travels in the file. ----
func fontVectors(ttf []byte, out string) error {
	f, err := truetype.Parse(ttf) // watermark.go:31
	if err != nil {
		return err
	}
	type vec struct {
		Text   string                   `json:"text"`
		Size   float64                  `json:"size"`
		W      int                      `json:"w"`
		H      int                      `json:"h"`
		Px     int                      `json:"px"`
		Py     int                      `json:"py"`
		Frame  string                   `json:"frame_rgba_hex"` // the frame after DrawString onto a constant frame: what the composite must give
		Fill   []int                    `json:"fill"`
		Col    []int                    `json:"col"`
		Glyphs []map[string]interface{} `json:"glyphs,omitempty"`
	}
	var vecs []vec
	for _, tc := range []struct {
		text string
		size float64
		w, h int
	}{{"© ImageProcessor", 36, 640, 120}, {"moire, offsets: fjord Avery", 17.5, 400, 60}, {"oo rr ss ee", 24, 300, 60}} {
		dst := image.NewRGBA(image.Rect(0, 0, tc.w, tc.h))
		fill := color.RGBA{37, 99, 180, 255}
		draw.Draw(dst, dst.Bounds(), image.NewUniform(fill), image.Point{}, draw.Src)
		col := color.RGBA{255, 255, 255, 127}
		c := freetype.NewContext() // watermark.go:98-104
		c.SetDPI(72)
		c.SetFont(f)
		c.SetFontSize(tc.size)
		c.SetClip(dst.Bounds())
		c.SetDst(dst)
		c.SetSrc(image.NewUniform(col))
		px, py := 10, tc.h-12
		if _, err := c.DrawString(tc.text, freetype.Pt(px, py)); err != nil { // watermark.go:151
			return err
		}
		vecs = append(vecs, vec{Text: tc.text, Size: tc.size, W: tc.w, H: tc.h, Px: px, Py: py, Frame: hex.EncodeToString(dst.Pix),
			Fill: []int{int(fill.R), int(fill.G), int(fill.B), int(fill.A)}, Col: []int{255, 255, 255, 127}})
	}
	b, _ := json.MarshalIndent(map[string]interface{}{"note": "freetype.Context.DrawString of golang/freetype on the given face; origin go",
		"ttf_hex": hex.EncodeToString(ttf), "vectors": vecs}, "", " ")
	return os.WriteFile(out, b, 0o644)
}

// ---- jpeg.Encode / image.Decode: streams and planes for small seeded frames (the pixels travel in the file) ----
func jpegVectors(out string) error {
	rng := rand.New(rand.NewSource(20261004))
	var vecs []map[string]interface{}
	for _, sz := range [][2]int{{48, 32}, {17, 9}, {64, 64}, {1, 1}, {33, 47}} {
		for _, q := range []int{85, 30, 100} {
			m := image.NewRGBA(image.Rect(0, 0, sz[0], sz[1]))
			for i := range m.Pix {
				m.Pix[i] = uint8(rng.Intn(256))
			}
			for i := 3; i < len(m.Pix); i += 4 {
				m.Pix[i] = 255
			}
			var buf bytes.Buffer
			if err := jpeg.Encode(&buf, m, &jpeg.Options{Quality: q}); err != nil { // resize.go:80
				return err
			}
			dec, err := jpeg.Decode(bytes.NewReader(buf.Bytes())) // image_processor.go:47
			if err != nil {
				return err
			}
			y := dec.(*image.YCbCr)
			vecs = append(vecs, map[string]interface{}{"w": sz[0], "h": sz[1], "quality": q, "rgba_hex": hex.EncodeToString(m.Pix),
				"stream_hex": hex.EncodeToString(buf.Bytes()), "ystride": y.YStride, "cstride": y.CStride,
				"y_hex": hex.EncodeToString(y.Y), "cb_hex": hex.EncodeToString(y.Cb), "cr_hex": hex.EncodeToString(y.Cr)})
		}
	}
	b, _ := json.MarshalIndent(map[string]interface{}{"note": "image/jpeg Encode and Decode of the Go standard library; origin go", "vectors": vecs}, "", " ")
	return os.WriteFile(out, b, 0o644)
}

func main() {
	in := flag.String("in", "../../tests/golden/kats.json", "committed known answers (inputs are taken from here)")
	out := flag.String("out", "../../tests/golden/kats_go.json", "the same cases with Go's answers")
	font := flag.String("font", "", "TrueType file for the DrawString vectors (default: the Go Regular face the reference embeds)")
	fontOut := flag.String("fontout", "../../tests/golden/font_go.json", "")
	jpegOut := flag.String("jpegout", "../../tests/golden/jpeg_go.json", "")
	flag.Parse()
	raw, err := os.ReadFile(*in)
	if err != nil {
		panic(err)
	}
	var k kats
	if err := json.Unmarshal(raw, &k); err != nil {
		panic(err)
	}
	res := kats{Note: "output of the Go libraries the reference uses (x/image v0.33.0, image/draw); origin go"}
	differ := 0
	for _, c := range k.Cases {
		a := answer(c)
		if a == nil {
			continue
		}
		if !reflect.DeepEqual(a, c["expect"]) {
			differ++
			fmt.Printf("DIFFERS from the committed expectation: %v %v\n", c["kind"], c["name"])
		}
		g := map[string]interface{}{}
		for key, v := range c {
			g[key] = v
		}
		g["expect"], g["origin"] = a, "go"
		res.Cases = append(res.Cases, g)
	}
	b, _ := json.Marshal(res)
	if err := os.WriteFile(*out, b, 0o644); err != nil {
		panic(err)
	}
	fmt.Printf("%d cases written to %s, %d differ from the committed (hand / model) answers\n", len(res.Cases), *out, differ)
	ttf := goregular.TTF // watermark.go:30
	if *font != "" {
		if ttf, err = os.ReadFile(*font); err != nil {
			panic(err)
		}
	}
	if err := fontVectors(ttf, *fontOut); err != nil {
		panic(err)
	}
	if err := jpegVectors(*jpegOut); err != nil {
		panic(err)
	}
}
