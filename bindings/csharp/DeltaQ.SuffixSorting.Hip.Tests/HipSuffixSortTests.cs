// HipSuffixSortTests.cs -- the reference's provider tests re-pointed at the HIP shim: the same inputs
// (test/DeltaQ.SuffixSorting.LibDivSufSort.Tests/LibDivSufSortTests.cs:66-148 -- the shruggy string, every file of
// test/assets, Random(63*13*63*13) buffers of the same sizes), the same acceptance check (strict SequenceCompareTo order
// between neighbours, :43-59) and, stronger than the reference can ask of itself, equality with LibDivSufSort's
// array entry for entry -- the suffix array under that order is unique.
// Source only: no dotnet SDK in the build image.  tests/test_gpu_parity.py runs the same cases through the C ABI.
using DeltaQ.SuffixSorting.Hip;
using DeltaQ.SuffixSorting.LibDivSufSort;
using System;
using System.Collections.Generic;
using System.IO;
using System.Linq;
using System.Text;
using Xunit;

namespace DeltaQ.Tests;

public sealed class HipSuffixSortTests
{
    private const int Seed = 63 * 13 * 63 * 13;

    private static byte[] RandomBytes(int size)
    {
        var bytes = new byte[size];
        new Random(Seed).NextBytes(bytes);
        return bytes;
    }

    private static void AssertIsTheSuffixArray(ReadOnlySpan<byte> text, ReadOnlySpan<int> sa)
    {
        Assert.Equal(text.Length, sa.Length);
        for (int i = 0; i + 1 < sa.Length; i++)
        {
            Assert.True(text[sa[i]..].SequenceCompareTo(text[sa[i + 1]..]) < 0, $"suffixes {i} and {i + 1} are out of order");
        }

        using var expected = new LibDivSufSort().Sort(text);
        Assert.True(expected.Memory.Span.SequenceEqual(sa), "differs from LibDivSufSort.Sort");
    }

    [Fact]
    public void Shruggy()
    {
        ReadOnlySpan<byte> text = Encoding.UTF8.GetBytes(@"¯\_(ツ)_/¯");
        using var owner = new HipSuffixSort().Sort(text);
        AssertIsTheSuffixArray(text, owner.Memory.Span);
        Assert.Equal(new[] { 4, 8, 10, 2, 3, 9, 6, 7, 12, 1, 11, 0, 5 }, owner.Memory.Span.ToArray());
    }

    // every file of test/assets, including the two the LibDivSufSort list leaves out
    public static IEnumerable<object[]> Assets => Directory.EnumerateFiles("assets").OrderBy(p => p).Select(p => new object[] { p });

    [Theory]
    [MemberData(nameof(Assets))]
    public void AssetFile(string path)
    {
        ReadOnlySpan<byte> text = File.ReadAllBytes(path);
        using var owner = new HipSuffixSort().Sort(text);
        AssertIsTheSuffixArray(text, owner.Memory.Span);
    }

    [Theory]
    [InlineData(0)]
    [InlineData(1)]
    [InlineData(2)]
    [InlineData(4)]
    [InlineData(8)]
    [InlineData(16)]
    [InlineData(32)]
    [InlineData(51)]
    [InlineData(0x1000)]
    [InlineData(0x8000)]
    [InlineData(0x8000 - 1)]
    [InlineData(1 << 20)]
    [InlineData((16 << 20) + 3)]
    public void RandomBuffer(int size)
    {
        byte[] text = RandomBytes(size);
        var sa = new int[size + 1];
        sa[size] = 12345;                                           // Diff.Create's sentinel slot (Diff.cs:78,89-90) stays untouched
        new HipSuffixSort().Sort(text, sa.AsSpan(0, size));
        AssertIsTheSuffixArray(text, sa.AsSpan(0, size));
        Assert.Equal(12345, sa[size]);
    }

    [Fact]
    public void LengthMismatchThrowsLikeLibDivSufSort()
    {
        var ex = Assert.Throws<ArgumentException>(() => new HipSuffixSort().Sort(new byte[4], new int[5]));
        Assert.Equal("Text and suffix buffers should have the same length", ex.Message);
    }

    [Fact]
    public void BadDeviceOrdinalIsACallerErrorEvenWithAFallback()
    {
        // device ordinal 63 does not exist: the native call returns DQ_ERR_BAD_ARGS (-1), which no fallback hides
        Assert.Throws<InvalidOperationException>(() => new HipSuffixSort(63, new LibDivSufSort()).Sort(new byte[100]).Dispose());
    }
}
