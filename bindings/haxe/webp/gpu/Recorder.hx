package webp.gpu;

import haxe.io.Bytes;
import haxe.io.BytesBuffer;

/**
    What `Vp8Decoder.reconstruct`'s parse half leaves behind for the device, in the packed format of
    include/webp_amd.h: one 16-byte record per macroblock, the non-zero 4x4 coefficient blocks (16 x int16 each, in the
    order WHT, luma 0..15, Cb 0..3, Cr 0..3) and -- once the whole frame is parsed, because FilterParam.inner is shared
    and sticky (Vp8Decoder.hx:348-353) -- one loop-filter record per macroblock.

    With `-D webp_gpu` the decoder calls `macroblock()` where it would have called prepareYBR / reconstructMacroblock /
    copy, and `finish()` where it would have run simpleFilter / normalFilter (patches/Vp8Decoder.hx.patch).
**/
@:access(webp.vp8.Vp8Decoder)
class Recorder {
    public var width:Int;
    public var height:Int;
    public var mbw:Int;
    public var mbh:Int;
    public var filterType:Int = 0;      // 0: filterHeader.level == 0, 1: simple, 2: normal
    public var mb:Bytes;                // 4 x uint32 per MB
    public var filt:Bytes;              // 1 x uint32 per MB
    public var nBlocks:Int = 0;
    var coeff:BytesBuffer;

    public function new(width:Int, height:Int, mbw:Int, mbh:Int) {
        this.width = width; this.height = height; this.mbw = mbw; this.mbh = mbh;
        mb = Bytes.alloc(16 * mbw * mbh);
        filt = Bytes.alloc(4 * mbw * mbh);
        coeff = new BytesBuffer();
    }

    /** Stores block `base`..`base+15` of dec.coeff if it holds a non-zero value (ignoring index 0 when skipDC). **/
    function block(dec:webp.vp8.Vp8Decoder, base:Int, skipDC:Bool):Bool {
        var any = 0;
        for (i in (skipDC ? 1 : 0)...16) any |= dec.coeff[base + i];
        if (any == 0) return false;
        for (i in 0...16) {
            final c = (skipDC && i == 0) ? 0 : dec.coeff[base + i];      // i16 luma blocks carry DC = 0: the device runs the WHT in int32 (SURVEY Q5)
            coeff.addByte(c & 0xff);
            coeff.addByte((c >> 8) & 0xff);
        }
        nBlocks++;
        return true;
    }

    /** Called from reconstruct() after parsePredMode* / parseResiduals, in place of prepareYBR + reconstructMacroblock + copy. **/
    public function macroblock(dec:webp.vp8.Vp8Decoder, mbx:Int, mby:Int):Void {
        final idx = mby * mbw + mbx;
        final first = nBlocks;
        var w0 = 0, w1 = 0, w2 = 0;
        if (dec.usePredY16) {
            w0 = dec.predY16;
            w2 |= 1 << 25;
            if (block(dec, 384, false)) w2 |= 1 << 24;                     // the 16 WHT inputs (whtCoeffBase)
        } else {
            for (j in 0...4) for (i in 0...4) {
                final n = 4 * j + i, p = dec.predY4[j][i];
                if (n < 8) w0 |= p << (4 * n); else w1 |= p << (4 * (n - 8));
            }
        }
        for (n in 0...16) if (block(dec, 16 * n, dec.usePredY16)) w2 |= 1 << n;
        for (n in 0...8) if (block(dec, 256 + 16 * n, false)) w2 |= 1 << (16 + n);   // Cb 256.., Cr 320..
        w2 |= dec.predC8 << 26;
        mb.setInt32(16 * idx, w0); mb.setInt32(16 * idx + 4, w1); mb.setInt32(16 * idx + 8, w2); mb.setInt32(16 * idx + 12, first);
    }

    /** Called from decodeFrame() after the macroblock loop, in place of simpleFilter() / normalFilter(). **/
    public function finish(dec:webp.vp8.Vp8Decoder):Void {
        filterType = dec.filterHeader.level == 0 ? 0 : (dec.filterHeader.simple ? 1 : 2);
        for (idx in 0...mbw * mbh) {
            final f = dec.perMBFilterParams[idx];                            // read now: `inner` is final (Q1)
            filt.setInt32(4 * idx, (f.level & 0xff) | (f.ilevel & 0xff) << 8 | (f.hlevel & 0xff) << 16 | (f.inner ? 1 : 0) << 24);
        }
    }

    public function coeffBytes():Bytes {
        final b = coeff.getBytes();
        coeff = new BytesBuffer();
        coeff.add(b);
        return b.length > 0 ? b : Bytes.alloc(32);
    }
}
