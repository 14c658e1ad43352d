package webp.gpu;

import haxe.io.Bytes;

/**
    Target-neutral front of the native backend: packs descriptors, calls Native (HashLink) or NativeCpp (hxcpp) and
    re-throws a failure as the String the reference's own code would throw (`throw "<String>"`, SURVEY 5).
    Descriptor layouts are those of include/webp_amd.h (little-endian, natural alignment).
**/
class Gpu {
    public static inline final FRAME_BYTES = 96;      // sizeof(wp_vp8_frame)
    static var ready = false;

    public static function ensure():Void {
        if (ready) return;
        check(#if hl Native.init(0) #elseif cpp NativeCpp.init(0) #else -7 #end);
        ready = true;
    }

    public static function lastError():String {
        final buf = Bytes.alloc(512);
        #if hl
        Native.last_error(@:privateAccess buf.b, 512);
        #elseif cpp
        NativeCpp.lastError(cast cpp.NativeArray.address(buf.getData(), 0).raw, 512);
        #end
        var n = 0;
        while (n < 512 && buf.get(n) != 0) n++;
        return buf.getString(0, n);
    }

    public static inline function check(rc:Int):Void {
        if (rc != 0) throw lastError();
    }

    /** wp_vp8_frame for one frame whose records start at index 0 of their arrays and whose outputs start at offset 0. **/
    public static function frameDesc(width:Int, height:Int, mbw:Int, mbh:Int, filterType:Int, hasAlpha:Bool):Bytes {
        final d = Bytes.alloc(FRAME_BYTES);
        d.fill(0, FRAME_BYTES, 0);
        d.setInt32(0, width); d.setInt32(4, height); d.setInt32(8, mbw); d.setInt32(12, mbh);
        d.setInt32(16, filterType); d.setInt32(20, hasAlpha ? 1 : 0);
        // mb_base, stile_base, cost / turn, coeff_blocks / flags (the host-pointer entry points plan a copy and bound the frame by the
        // coefficient count they are given), coeff_base and all offsets stay 0
        return d;
    }

    /** recon + loop filter of one recorded frame: the Yuv420 planes Vp8Decoder.decodeFrame returns (MB-padded). **/
    public static function decodeFrame(rec:Recorder):{Y:Bytes, Cb:Bytes, Cr:Bytes, YStride:Int, CStride:Int} {
        ensure();
        final nmb = rec.mbw * rec.mbh;
        final y = Bytes.alloc(256 * nmb), u = Bytes.alloc(64 * nmb), v = Bytes.alloc(64 * nmb);
        final desc = frameDesc(rec.width, rec.height, rec.mbw, rec.mbh, rec.filterType, false);
        final coeff = rec.coeffBytes();
        #if hl
        check(Native.vp8_decode_to_argb(1, @:privateAccess desc.b, @:privateAccess rec.mb.b, nmb, @:privateAccess rec.filt.b,
            @:privateAccess coeff.b, rec.nBlocks, @:privateAccess y.b, @:privateAccess u.b, @:privateAccess v.b, null, null));
        #elseif cpp
        inline function p(b:Bytes) return cpp.NativeArray.address(b.getData(), 0).raw;
        check(NativeCpp.vp8DecodeToArgb(1, cast p(desc), cast p(rec.mb), nmb, cast p(rec.filt), cast p(coeff), rec.nBlocks,
            cast p(y), cast p(u), cast p(v), null, null));
        #end
        return {Y: y, Cb: u, Cr: v, YStride: 16 * rec.mbw, CStride: 8 * rec.mbw};
    }

    /** Tools.toArgb's Yuv420 case (Tools.hx:10-44): tightly packed A,R,G,B. **/
    public static function toArgb(width:Int, height:Int, y:Bytes, ystride:Int, u:Bytes, v:Bytes, ?a:Bytes):Bytes {
        ensure();
        final mbw = ystride >> 4, mbh = Std.int(y.length / (ystride * 16));
        final desc = frameDesc(width, height, mbw, mbh, 0, a != null);
        final out = Bytes.alloc(4 * width * height);
        #if hl
        check(Native.yuv420_to_argb(1, @:privateAccess desc.b, @:privateAccess y.b, @:privateAccess u.b, @:privateAccess v.b,
            a == null ? null : @:privateAccess a.b, @:privateAccess out.b));
        #elseif cpp
        inline function p(b:Bytes) return cpp.NativeArray.address(b.getData(), 0).raw;
        check(NativeCpp.yuv420ToArgb(1, cast p(desc), cast p(y), cast p(u), cast p(v), a == null ? null : cast p(a), cast p(out)));
        #end
        return out;
    }
}
